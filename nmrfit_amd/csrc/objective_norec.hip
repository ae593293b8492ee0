// objective_norec.hip -- the instantiations of objective_kernel for NMRFIT_VARIANT_NOREC (objective and residual launches,
// imaginary-channel modes, four- and eight-wave workgroups): a translation unit of its own so that the variants
// compile in parallel.
#include "objective_kernel.h"

namespace nmrfit {
int launch_objective_norec(const ObjectiveLaunch &a) { return launch_variant<NMRFIT_VARIANT_NOREC>(a); }
}  // namespace nmrfit
