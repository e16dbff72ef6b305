// tu_pair16.hip - translation unit of its own so that the instantiation families compile side by side: conv_sx_pair16_kernel (two dependent ResBlock convs per launch on the 16x16x32 loop) + launcher
#define VITSMI_TU 1
#define VITSMI_IMPL_PAIR16 1
#include "conv_sx_pair16.hip.hpp"
