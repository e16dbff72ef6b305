// tu_sx_bf16.hip - translation unit of its own so that the instantiation families compile side by side: conv_sx_kernel, bf16 plane arithmetics (bf16x6 exact mode)
#define VITSMI_TU 1
#define VITSMI_IMPL_SX_BF16 1
#include "conv_sx_engine.hip.hpp"
