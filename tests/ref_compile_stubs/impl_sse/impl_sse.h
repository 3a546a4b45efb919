/* redirect: where the reference's hmmer.h:1048 includes the SSE implementation header, the drop-in's header is included instead */
#include "impl_hip.h"
