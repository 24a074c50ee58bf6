// libextrack_hip.so: register-resident 2-state kernels (xt_reg2.h), frame_len 7.  See xt_reg2_inst.h.
#define XT_R2_F 7
#include "xt_reg2_inst.h"
