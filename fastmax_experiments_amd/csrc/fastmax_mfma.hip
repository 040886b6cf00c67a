// placeholder: matrix-core kernel lands here
#include "fastmax_common.h"
namespace fastmax {
bool mfma_p1_supported(const fastmax_problem&) { return false; }
size_t mfma_p1_workspace(const fastmax_problem&) { return 0; }
int launch_fwd_mfma_p1(const FwdArgs&) { return FASTMAX_E_BAD_SHAPE; }
}
