// tu_sx_s32.hip - translation unit of its own so that the instantiation families compile side by side: conv_sx_kernel, f16x3 arithmetic on the 32x32x16 main loop (incl. the raw-input kernels)
#define VITSMI_TU 1
#define VITSMI_IMPL_SX_S32 1
#include "conv_sx_engine.hip.hpp"
