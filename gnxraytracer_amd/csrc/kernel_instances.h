// kernel_instances.h -- the instantiations of the two largest kernel templates, k_whitted_step and k_vol_step, are compiled in
// translation units of their own (inst_whitted.hip, inst_whitted_tex.hip, inst_vol.hip) so that the device compilations run side by side;
// api.hip sees them as `extern template`.  X(...) receives the template arguments of one instantiation.
#pragma once
#include <hip/hip_runtime.h>

#include "host_scene.h"
#include "kernels.hip.h"
#include "vol_kernel.hip.h"
#include "whitted_kernel.hip.h"

#define GX_WH_SPH(X, M, L, T) X(M, L, false, T) X(M, L, true, T)
#define GX_WH_LT(X, M, T) GX_WH_SPH(X, M, LT_AREA, T) GX_WH_SPH(X, M, LT_ALL, T)
#define GX_WHITTED_INSTANCES_TEX(X, T) GX_WH_LT(X, WM_WHITTED, T) GX_WH_LT(X, WM_DIRECT_ONE, T) GX_WH_LT(X, WM_DIRECT_ALL, T)
#define GX_WHITTED_INSTANCES(X) GX_WHITTED_INSTANCES_TEX(X, false) GX_WHITTED_INSTANCES_TEX(X, true)

#define GX_VS_ST(X, M, L, T) X(M, L, VS_MAIN, T) X(M, L, VS_SHADOW, T) X(M, L, VS_MIS, T)
#define GX_VS_LT(X, M, T) GX_VS_ST(X, M, LT_AREA, T) GX_VS_ST(X, M, LT_ALL, T)
#define GX_VOL_INSTANCES(X) GX_VS_LT(X, LM_DIFFUSE, false) GX_VS_LT(X, LM_GLOSSY, false) GX_VS_LT(X, LM_ALL, false) GX_VS_LT(X, LM_ALL, true)

#define GX_WHITTED_SIGNATURE(M, L, S, T) \
    __global__ void gnxr::k_whitted_step<M, L, S, T>(gnxr::DScene, gnxr::DRender, gnxr::PathArrays, gnxr::WhittedArrays, const int *, int, unsigned long long *);
#define GX_VOL_SIGNATURE(M, L, ST, T) \
    __global__ void gnxr::k_vol_step<M, L, ST, T>(gnxr::DScene, gnxr::DMediaTables, gnxr::DRender, gnxr::PathArrays, gnxr::VolArrays, const int *, const unsigned int *, int, int);
