// inst_whitted.hip -- explicit instantiations of k_whitted_step (Whitted + DirectLighting state machine) for scenes without
// image textures / per-corner uvs, see kernel_instances.h
#include "kernel_instances.h"
using namespace gnxr;
#define X(M, L, S, T) template GX_WHITTED_SIGNATURE(M, L, S, T)
GX_WHITTED_INSTANCES_TEX(X, false)
#undef X
