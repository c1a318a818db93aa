// inst_whitted_tex.hip -- explicit instantiations of k_whitted_step with ray differentials and image textures (TEX), see
// kernel_instances.h
#include "kernel_instances.h"
using namespace gnxr;
#define X(M, L, S, T) template GX_WHITTED_SIGNATURE(M, L, S, T)
GX_WHITTED_INSTANCES_TEX(X, true)
#undef X
