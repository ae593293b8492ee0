// objective_farfield32.hip -- NMRFIT_VARIANT_FARFIELD32: the far-field kernel with orders 1..15 of its shared polynomial in
// packed fp32 (opt-in mixed precision; objective launches without the imaginary channel, four- and eight-wave workgroups).
// Residual rows and the imaginary channel of a context set to this variant run the fp64 far-field kernel.
#include "objective_kernel.h"

namespace nmrfit {
int launch_objective_farfield32(const ObjectiveLaunch &a) { return launch_variant<NMRFIT_VARIANT_FARFIELD32>(a); }
}  // namespace nmrfit
