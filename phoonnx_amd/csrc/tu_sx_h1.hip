// tu_sx_h1.hip - translation unit of its own so that the instantiation families compile side by side: conv_sx_kernel, one fp16 plane / one product (BASELINE config 4) on the 16x16x32 main loop
#define VITSMI_TU 1
#define VITSMI_IMPL_SX_H1 1
#include "conv_sx_engine.hip.hpp"
