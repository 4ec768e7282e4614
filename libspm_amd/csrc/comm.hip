// comm.hip -- the multi-GPU exchange step (RCCL gatherv of hit records).
// MI355X only; no CPU scan path exists in this library: if HIP fails the call fails.
#include "internal.hpp"
#include "comm.hpp"
