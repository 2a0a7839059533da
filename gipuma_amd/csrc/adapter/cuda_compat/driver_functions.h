/* include-name forwarder: the reference includes <driver_functions.h>; on MI355X that is gipuma_cuda_compat.h */
#include "gipuma_cuda_compat.h"
