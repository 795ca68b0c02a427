// kernels_conv_bf16.hip — the plain-bf16 instantiations (NP = 1: one bf16 plane, one MFMA per product) of the bf16-matrix-core
// layers, HNET_PREC_BF16.  A REPORTED arithmetic mode (BASELINE.json config 2 names "bf16"; SURVEY.md fact 5: ~1e-2 px,
// outside the 1e-4 px parity gate): the same kernels and dispatch as the split-bf16 default (s3_dispatch.h), compiled in
// their own translation unit so that the build stays parallel.
#include "s3_dispatch.h"

namespace hnet {

HNET_S3_DISPATCH_INSTANCES(, 1)

}  // namespace hnet
