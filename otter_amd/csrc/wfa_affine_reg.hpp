// wfa_affine_reg.hpp — host-side entry of the register-resident gap-affine tiers (wfa_affine_reg.hip), called by the launch chain in
// wfa_affine.hip.  Tier = window: 0 = 1024 diagonals (one wave x 8 pair-slots), 1 = 1536 (1 x 12), 2 = 2048 (1 x 16), 3 = 4096 (4 waves x 8),
// 4 = 8192 (8 x 8).
#pragma once
#include "wfa_affine_common.hpp"

constexpr int OTG_REG_TIERS = 5;
// diagonals of window / bytes of packed sequence pair a tier admits (the counting sort in wfa_affine.hip and the kernels use the same figures)
constexpr int OTG_REG_CAP[OTG_REG_TIERS] = {1024, 1536, 2048, 4096, 8192};
constexpr int OTG_REG_SEQB[OTG_REG_TIERS] = {4096, 4608, 6144, 8192, 12288};
// Geometry of a tier's launch: alignments per block and resident blocks per CU (one-wave shapes: four alignments per 256-thread block;
// multi-wave shapes: one alignment per block of NW waves).  `shape` selects among the instantiations of a window (0 = the chain's default;
// the others are measurement switches, OTG_REG_SHAPE = one digit per tier).
void otg_affine_reg_geometry(int tier, int shape, int* aln_per_block, int* blocks_per_cu);

int otg_launch_affine_reg_tier(otg_ctx* ctx, int tier, int shape, uint32_t blocks, const uint8_t* d_arena, const otg_align_task* d_tasks,
                               const uint32_t* d_sorted, const uint32_t* d_seg, int g, int32_t* d_scores, const uint64_t* d_cig_off,
                               uint32_t* d_cig_len, uint8_t* d_cig_arena, uint64_t* d_cells, uint32_t* ticket, uint32_t* n_overflow,
                               uint32_t* overflow_list, const otg_affine::AffWs& ws, const int32_t* d_bound, unsigned long long* visited);
