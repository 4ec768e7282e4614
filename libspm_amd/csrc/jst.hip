// jst.hip -- journaled-sequence (pan-genome) search, config C5.
// MI355X only; no CPU scan path exists in this library: if HIP fails the call fails.
#include "internal.hpp"
#include "synth.hpp"
#include "jst.hpp"
