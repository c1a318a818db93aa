// ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h header).
// Result of <Light>::Sample_Li (core/Light.h:40-43): radiance, direction, pdf and the
// VisibilityTester end point p1.
#pragma once
#include "o_scene.h"
namespace gnxo {
struct LightSample {
    Spec Li;
    V3 wi;
    Float pdf = 0;
    Interaction p1;
};
}  // namespace gnxo
