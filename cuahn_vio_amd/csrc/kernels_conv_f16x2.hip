// kernels_conv_f16x2.hip — the fp16-plane instantiations (NP = 2: two fp16 planes per activation, three fp16 MFMAs per product;
// s3_format.h) of the matrix-core layers, HNET_PREC_F16X2.  fp32-grade like the split-bf16 default (same parity gates) at half its
// matrix-core work and two thirds of its activation bytes.  Same kernels and dispatch (s3_dispatch.h), compiled in their own
// translation unit so that the build stays parallel.
#include "s3_dispatch.h"

namespace hnet {

HNET_S3_DISPATCH_INSTANCES(, 2)

}  // namespace hnet
