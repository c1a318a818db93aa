// ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h header).
// o_media.h: VolPathIntegrator / WhittedIntegrator restatements (filled in after the Path slice).
#pragma once
#include "o_integrator.h"
namespace gnxo {
inline Spec VolPathLi(const RenderContext &, const PathParams &, const Ray &, SampleStream &) { return Spec(0.f); }
inline Spec WhittedLi(const RenderContext &, const PathParams &, const Ray &, SampleStream &, int) { return Spec(0.f); }
}  // namespace gnxo
