#include "otg_common.hpp"
void otg_pipeline_free(otg_ctx*) {}
