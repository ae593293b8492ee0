// objective_ab.hip -- the A/B forms of the objective kernel (BASELINE: IEEE divide + libdevice exp2 per unit, NOSKIP,
// SINGLE, QUAD, STAGED): what the tuned kernels are measured and checked against (tools/ab.py, bench.py's `variants`
// entry, the parity tests).  Compiled only into -DNMRFIT_AB_BUILD libraries (libnmrfit_amd_ab.so); the product library
// answers NMRFIT_E_UNSUPPORTED for these variants.
#ifdef NMRFIT_AB_BUILD
#include "objective_kernel.h"

namespace nmrfit {
int launch_objective_ab(int variant, const ObjectiveLaunch &a)
{
    switch (variant) {
        case NMRFIT_VARIANT_BASELINE: return launch_variant<NMRFIT_VARIANT_BASELINE>(a);
        case NMRFIT_VARIANT_NOSKIP: return launch_variant<NMRFIT_VARIANT_NOSKIP>(a);
        case NMRFIT_VARIANT_SINGLE: return launch_variant<NMRFIT_VARIANT_SINGLE>(a);
        case NMRFIT_VARIANT_QUAD: return launch_variant<NMRFIT_VARIANT_QUAD>(a);
        case NMRFIT_VARIANT_STAGED: return launch_variant<NMRFIT_VARIANT_STAGED>(a);
        default: break;
    }
    set_error("not an A/B variant");
    return NMRFIT_E_INVALID;
}
}  // namespace nmrfit
#endif
