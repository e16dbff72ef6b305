// tu_sx.hip - translation unit of its own so that the instantiation families compile side by side: launch_conv_sx: argument preparation and dispatch to the instantiation families
#define VITSMI_TU 1
#define VITSMI_IMPL_SX 1
#include "conv_sx_engine.hip.hpp"
