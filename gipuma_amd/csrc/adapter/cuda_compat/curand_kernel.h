/* include-name forwarder: the reference includes <curand_kernel.h>; on MI355X that is gipuma_cuda_compat.h */
#include "gipuma_cuda_compat.h"
