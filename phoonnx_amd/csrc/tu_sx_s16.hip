// tu_sx_s16.hip - translation unit of its own so that the instantiation families compile side by side: conv_sx_kernel, f16x3 arithmetic on the 16x16x32 main loop
#define VITSMI_TU 1
#define VITSMI_IMPL_SX_S16 1
#include "conv_sx_engine.hip.hpp"
