// tu_conv_f32.hip - translation unit of its own so that the instantiation families compile side by side: the f32 engine (conv_engine_kernel instantiations + launch_conv)
#define VITSMI_TU 1
#define VITSMI_IMPL_CONV_F32 1
#include "conv_engine.hip.hpp"
