// host_model.hpp -- shared host-side helpers (see host_model.cpp).
#pragma once
#include "bath_hip.h"

namespace bath {
extern const float kNegInf;
extern const float kAminoBg[20];
bool amino_degen_has(int x, int y);
void core_transitions(const bath_hmm &h, float *tsc);
void length_model(float xsc[4][2], float nj, int L);
void match_logodds(const bath_hmm &h, int k, float sc[BATH_KP_AMINO]);
}  // namespace bath
