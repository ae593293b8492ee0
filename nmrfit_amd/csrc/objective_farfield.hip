// objective_farfield.hip -- the instantiations of objective_kernel for NMRFIT_VARIANT_FARFIELD (objective and residual launches,
// imaginary-channel modes, four- and eight-wave workgroups): a translation unit of its own so that the variants
// compile in parallel.
#include "objective_kernel.h"

namespace nmrfit {
int launch_objective_farfield(const ObjectiveLaunch &a) { return launch_variant<NMRFIT_VARIANT_FARFIELD>(a); }
}  // namespace nmrfit
