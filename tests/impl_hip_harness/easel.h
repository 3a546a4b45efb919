/* TEST HARNESS ONLY: the harness keeps every easel declaration impl_hip names in its one header */
#include "hmmer.h"
