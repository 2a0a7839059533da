/* include-name forwarder: the reference includes <cuda_gl_interop.h>; on MI355X that is gipuma_cuda_compat.h */
#include "gipuma_cuda_compat.h"
