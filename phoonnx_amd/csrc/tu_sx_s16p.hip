// tu_sx_s16p.hip - translation unit of its own so that the instantiation families compile side by side: conv_sx_kernel, f16x3 on the 16x16x32 loop with the residual read from operand planes (SX_RES_PL)
#define VITSMI_TU 1
#define VITSMI_IMPL_SX_S16P 1
#include "conv_sx_engine.hip.hpp"
