#include "otg_common.hpp"
