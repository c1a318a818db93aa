// inst_vol.hip -- explicit instantiations of k_vol_step (VolPath state machine), see kernel_instances.h
#include "kernel_instances.h"
using namespace gnxr;
#define X(M, L, ST, T) template GX_VOL_SIGNATURE(M, L, ST, T)
GX_VOL_INSTANCES(X)
#undef X
