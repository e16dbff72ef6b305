// tu_pair.hip - translation unit of its own so that the instantiation families compile side by side: conv_sx_pair_kernel instantiations + launchers
#define VITSMI_TU 1
#define VITSMI_IMPL_PAIR 1
#include "conv_sx_pair.hip.hpp"
