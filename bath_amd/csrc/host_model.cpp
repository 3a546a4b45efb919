// host_model.cpp -- host-side model objects of the bath_hip library (no GPU code here).
//
// Mirrors the generic layer the reference's impl boundary consumes:
//   bath_hmmfile_read      <- read_asc30hmm()            src/p7_hmmfile.c:1342-1697
//   bath_profile_config    <- p7_ProfileConfig()         src/modelconfig.c:48-196   (p7_LOCAL, multihit)
//   bath_fs_profile_config <- p7_ProfileConfig_fs()      src/modelconfig.c:220-698
//   bath_gencode_basic     <- esl_gencode_Set()/basic[]  (easel; used at modelconfig.c:364)
// NB the reference calls C's double log() on float arguments; std::log(float) would pick logf, hence the casts.
// Written from scratch in C++; numerics (float/double promotion order) follow the cited lines so the
// resulting score tables are bit-identical to the reference's.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "bath_hip.h"
#include "host_model.hpp"

namespace bath {

const float kNegInf = -std::numeric_limits<float>::infinity();

// Swiss-Prot 50.8 amino acid background, p7_AminoFrequencies() src/hmmer.c:161-184
const float kAminoBg[20] = {
    0.0787945f, 0.0151600f, 0.0535222f, 0.0668298f, 0.0397062f, 0.0695071f, 0.0229198f, 0.0590092f,
    0.0594422f, 0.0963728f, 0.0237718f, 0.0414386f, 0.0482904f, 0.0395639f, 0.0540978f, 0.0683364f,
    0.0540687f, 0.0673417f, 0.0114135f, 0.0304133f};

static const char kAminoSyms[] = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~";

// Members of each degenerate amino code (easel alphabet: B=DN J=IL Z=EQ O=K U=C X=all).
bool amino_degen_has(int x, int y) {
  if (x < 20) return x == y;
  switch (x) {
    case 21: return y == 2 || y == 11;
    case 22: return y == 7 || y == 9;
    case 23: return y == 3 || y == 13;
    case 24: return y == 8;
    case 25: return y == 1;
    case 26: return y >= 0 && y < 20;
    default: return false;
  }
}

// esl_abc_FExpectScVec: degenerate residue score = background-weighted mean over its members.
static void fill_degenerate_scores(float *sc) {
  for (int x = 21; x <= 26; x++) {
    float num = 0.f, den = 0.f;
    for (int y = 0; y < 20; y++)
      if (amino_degen_has(x, y)) { num += sc[y] * kAminoBg[y]; den += kAminoBg[y]; }
    sc[x] = num / den;
  }
}

static inline float prob_from_token(const std::string &tok) {
  if (!tok.empty() && tok[0] == '*') return 0.0f;
  return expf((float)(-1.0 * atof(tok.c_str())));   // p7_hmmfile.c:1600
}

}  // namespace bath

using namespace bath;

extern "C" int bath_hmmfile_count(const char *path) {
  std::ifstream in(path);
  if (!in) return -1;
  std::string line; int n = 0;
  while (std::getline(in, line)) if (line.compare(0, 2, "//") == 0) n++;
  return n;
}

extern "C" void bath_hmm_destroy(bath_hmm *h) {
  if (!h) return;
  delete[] h->t; delete[] h->mat; delete[] h->ins; delete[] h->consensus; delete[] h->rf; delete[] h->cs; delete h;
}

extern "C" int bath_hmmfile_read(const char *path, int index, bath_hmm **ret) {
  *ret = nullptr;
  std::ifstream in(path);
  if (!in) return BATH_EFAIL;
  std::string line;
  for (int seen = 0; seen < index;) {
    if (!std::getline(in, line)) return BATH_EFORMAT;
    if (line.compare(0, 2, "//") == 0) seen++;
  }
  if (!std::getline(in, line)) return BATH_EFORMAT;
  if (line.compare(0, 7, "BATH3/f") != 0 && line.compare(0, 8, "HMMER3/f") != 0) return BATH_EFORMAT;

  bath_hmm *h = new bath_hmm();
  std::memset(h, 0, sizeof *h);
  h->ct = 1; h->fsprob = 0.01f;
  for (float &e : h->evparam) e = -99999.0f;
  auto fail = [&](int code) { bath_hmm_destroy(h); return code; };

  bool body = false, has_cons = false, has_rf = false, has_cs = false;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string tag; ss >> tag;
    if (tag == "NAME") { std::string v; ss >> v; std::snprintf(h->name, sizeof h->name, "%s", v.c_str()); }
    else if (tag == "ACC") { std::string v; ss >> v; std::snprintf(h->acc, sizeof h->acc, "%s", v.c_str()); }
    else if (tag == "CONS") { std::string v; ss >> v; has_cons = (v == "yes"); }
    else if (tag == "RF") { std::string v; ss >> v; has_rf = (v == "yes"); }
    else if (tag == "CS") { std::string v; ss >> v; has_cs = (v == "yes"); }
    else if (tag == "LENG") ss >> h->M;
    else if (tag == "MAXL") ss >> h->max_length;
    else if (tag == "ALPH") { std::string v; ss >> v; if (v != "amino") return fail(BATH_EFORMAT); }
    else if (tag == "STATS") {
      std::string a, b, c, d; ss >> a >> b >> c >> d;
      if (d.empty()) return fail(BATH_EFORMAT);
      if      (b == "MSV")     { h->evparam[0] = (float)atof(c.c_str()); h->evparam[1] = (float)atof(d.c_str()); }
      else if (b == "VITERBI") { h->evparam[2] = (float)atof(c.c_str()); h->evparam[3] = (float)atof(d.c_str()); }
      else if (b == "FORWARD") { h->evparam[4] = (float)atof(c.c_str()); h->evparam[5] = (float)atof(d.c_str()); }
      else if (b == "FS3")     { h->evparam[6] = (float)atof(d.c_str()); }     // "STATS LOCAL FS3 FORWARD tau lambda": 4th token is tau (p7_hmmfile.c:1509)
      else if (b == "FS5")     { h->evparam[7] = (float)atof(d.c_str()); }
    }
    else if (tag == "FRAMESHIFT") { std::string a, v; ss >> a >> v; h->fsprob = (float)atof(v.c_str()); }
    else if (tag == "CODON")      { std::string a, v; ss >> a >> v; h->ct = atoi(v.c_str()); }
    else if (tag == "HMM") { body = true; break; }
  }
  if (!body || h->M <= 0) return fail(BATH_EFORMAT);
  if (!std::getline(in, line)) return fail(BATH_EFORMAT);   // transition header line

  const int M = h->M;
  h->t   = new float[(size_t)(M + 1) * 7]();
  h->mat = new float[(size_t)(M + 1) * 20]();
  h->ins = new float[(size_t)(M + 1) * 20]();
  h->consensus = new char[(size_t)M + 2]();
  h->consensus[0] = ' ';
  if (has_rf) { h->rf = new char[(size_t)M + 2](); h->rf[0] = ' '; }      // p7_hmmfile.c:1629-1640: columns after the emissions are MAP CONS RF MM CS
  if (has_cs) { h->cs = new char[(size_t)M + 2](); h->cs[0] = ' '; }

  auto read_tokens = [&](std::vector<std::string> &toks) -> bool {
    if (!std::getline(in, line)) return false;
    toks.clear();
    std::istringstream ss(line);
    std::string t;
    while (ss >> t) toks.push_back(t);
    return true;
  };
  std::vector<std::string> tk;
  if (!read_tokens(tk)) return fail(BATH_EFORMAT);
  if (!tk.empty() && tk[0] == "COMPO") {
    if (tk.size() < 21) return fail(BATH_EFORMAT);
    for (int x = 0; x < 20; x++) h->compo[x] = prob_from_token(tk[1 + x]);
    if (!read_tokens(tk)) return fail(BATH_EFORMAT);
  }
  if (tk.size() < 20) return fail(BATH_EFORMAT);
  for (int x = 0; x < 20; x++) h->ins[x] = prob_from_token(tk[x]);
  if (!read_tokens(tk) || tk.size() < 7) return fail(BATH_EFORMAT);
  for (int x = 0; x < 7; x++) h->t[x] = prob_from_token(tk[x]);
  for (int k = 1; k <= M; k++) {
    if (!read_tokens(tk) || tk.size() < 21 || atoi(tk[0].c_str()) != k) return fail(BATH_EFORMAT);
    for (int x = 0; x < 20; x++) h->mat[(size_t)k * 20 + x] = prob_from_token(tk[1 + x]);
    if (has_rf) h->rf[k] = tk.size() >= 24 ? tk[23][0] : '-';
    if (has_cs) h->cs[k] = tk.size() >= 26 ? tk[25][0] : '-';
    if (has_cons && tk.size() >= 23) h->consensus[k] = tk[22][0];             // columns after the emissions: MAP CONS RF MM CS (p7_hmmfile.c:1624-1640)
    else {                                                                    // p7_hmm_SetConsensus for an amino model
      int best = 0;
      for (int x = 1; x < 20; x++) if (h->mat[(size_t)k * 20 + x] > h->mat[(size_t)k * 20 + best]) best = x;
      const char c = kAminoSyms[best];
      h->consensus[k] = h->mat[(size_t)k * 20 + best] >= 0.5f ? c : (char)std::tolower(c);
    }
    if (!read_tokens(tk) || tk.size() < 20) return fail(BATH_EFORMAT);
    for (int x = 0; x < 20; x++) h->ins[(size_t)k * 20 + x] = prob_from_token(tk[x]);
    if (!read_tokens(tk) || tk.size() < 7) return fail(BATH_EFORMAT);
    for (int x = 0; x < 7; x++) h->t[(size_t)k * 7 + x] = prob_from_token(tk[x]);
  }
  if (!std::getline(in, line) || line.compare(0, 2, "//") != 0) return fail(BATH_EFORMAT);
  *ret = h;
  return BATH_OK;
}

// NCBI genetic codes in NCBI's own TCAG order; rearranged into easel's ACGT order below.
static const char *ncbi_code(int id) {
  switch (id) {
    case 1: case 11: return "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    case 2:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSS**VVVVAAAADDEEGGGG";
    case 3:  return "FFLLSSSSYY**CCWWTTTTPPPPHHQQRRRRIIMMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    case 4:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    case 5:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSSSVVVVAAAADDEEGGGG";
    case 6:  return "FFLLSSSSYYQQCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    case 9:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG";
    case 10: return "FFLLSSSSYY**CCCWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    case 12: return "FFLLSSSSYY**CC*WLLLSPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    case 13: return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSGGVVVVAAAADDEEGGGG";
    case 14: return "FFLLSSSSYYY*CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG";
    case 16: return "FFLLSSSSYY*LCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    case 21: return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNNKSSSSVVVVAAAADDEEGGGG";
    case 22: return "FFLLSS*SYY*LCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    case 23: return "FF*LSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    case 24: return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSSKVVVVAAAADDEEGGGG";
    case 25: return "FFLLSSSSYY**CCGWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    default: return nullptr;
  }
}

// Start codons of the NCBI tables (gc.prt "sncbieaa", 'M' = may initiate), TCAG order: what esl_gencode_Set leaves in
// gcode->is_initiator and bathsearch keeps under -M only (bathsearch.c:718-719).  Easel is not in the reference tree: unpinned.
static const char *ncbi_starts(int id) {
  switch (id) {
    case 1:  return "---M---------------M---------------M----------------------------";
    case 2:  return "--------------------------------MMMM---------------M------------";
    case 3:  return "----------------------------------MM----------------------------";
    case 4:  return "--MM---------------M------------MMMM---------------M------------";
    case 5:  return "---M----------------------------MMMM---------------M------------";
    case 6:  return "-----------------------------------M----------------------------";
    case 9:  return "-----------------------------------M---------------M------------";
    case 10: return "-----------------------------------M----------------------------";
    case 11: return "---M---------------M------------MMMM---------------M------------";
    case 12: return "-------------------M---------------M----------------------------";
    case 13: return "---M------------------------------MM---------------M------------";
    case 14: return "-----------------------------------M----------------------------";
    case 16: return "-----------------------------------M----------------------------";
    case 21: return "-----------------------------------M---------------M------------";
    case 22: return "-----------------------------------M----------------------------";
    case 23: return "--------------------------------M--M---------------M------------";
    case 24: return "---M---------------M---------------M---------------M------------";
    case 25: return "---M-------------------------------M---------------M------------";
    default: return nullptr;
  }
}

extern "C" int bath_gencode_initiators(int ncbi_table, int initiator, uint8_t is_init[64]) {
  if (!is_init) return BATH_EINVAL;
  if (initiator == BATH_INIT_ANY) { std::memset(is_init, 1, 64); return BATH_OK; }     // esl_gencode_SetInitiatorAny (a stop still ends the ORF)
  std::memset(is_init, 0, 64);
  if (initiator == BATH_INIT_AUG) { is_init[16 * 0 + 4 * 3 + 2] = 1; return BATH_OK; } // esl_gencode_SetInitiatorOnlyAUG
  if (initiator != BATH_INIT_TABLE) return BATH_EINVAL;
  const char *st = ncbi_starts(ncbi_table);
  if (!st) return BATH_EINVAL;
  const int to_easel[4] = {3, 1, 0, 2};
  for (int i = 0; i < 64; i++) is_init[16 * to_easel[i >> 4] + 4 * to_easel[(i >> 2) & 3] + to_easel[i & 3]] = (uint8_t)(st[i] == 'M');
  return BATH_OK;
}

extern "C" int bath_gencode_basic(int ncbi_table, uint8_t basic[64]) {
  const char *code = ncbi_code(ncbi_table);
  if (!code) return BATH_EINVAL;
  const int to_easel[4] = {3, 1, 0, 2};   // T,C,A,G -> easel digital codes (A=0 C=1 G=2 T=3)
  for (int i = 0; i < 64; i++) {
    int a = to_easel[i >> 4], b = to_easel[(i >> 2) & 3], c = to_easel[i & 3];
    const char *p = std::strchr(kAminoSyms, code[i]);
    basic[16 * a + 4 * b + c] = (uint8_t)(p - kAminoSyms);
  }
  return BATH_OK;
}

namespace bath {

// Match-state occupancy, p7_hmm_CalculateOccupancy() src/p7_hmm.c:1348
static std::vector<float> match_occupancy(const bath_hmm &h) {
  std::vector<float> occ(h.M + 1, 0.f);
  occ[1] = h.t[1] + h.t[0];   // tMI + tMM at node 0
  for (int k = 2; k <= h.M; k++) {
    const float *tp = h.t + (size_t)(k - 1) * 7;
    occ[k] = (float)(occ[k - 1] * (tp[0] + tp[1]) + (1.0 - occ[k - 1]) * tp[5]);
  }
  return occ;
}

// Core transition scores shared by p7_ProfileConfig and p7_ProfileConfig_fs (modelconfig.c:86-136).
void core_transitions(const bath_hmm &h, float *tsc) {
  const int M = h.M;
  std::fill(tsc, tsc + (size_t)M * 8, kNegInf);
  std::vector<float> occ = match_occupancy(h);
  float Z = 0.f;
  for (int k = 1; k <= M; k++) Z += occ[k] * (float)(M - k + 1);
  for (int k = 1; k <= M; k++) tsc[(size_t)(k - 1) * 8 + 3] = (float)std::log((double)(occ[k] / Z));   // BM, stored off by one
  for (int k = 1; k < M; k++) {
    const float *t = h.t + (size_t)k * 7;   // MM MI MD IM II DM DD
    float *o = tsc + (size_t)k * 8;         // MM IM DM BM MD DD MI II
    o[0] = (float)std::log((double)t[0]);
    o[6] = (float)std::log((double)t[1]);
    o[4] = (float)std::log((double)t[2]);
    o[1] = (float)std::log((double)t[3]);
    o[7] = (float)std::log((double)t[4]);
    o[2] = (float)std::log((double)t[5]);
    o[5] = (float)std::log((double)t[6]);
  }
}

// N/C/J loop and move scores for expected length L (modelconfig.c:730-733).
void length_model(float xsc[4][2], float nj, int L) {
  float pmove = (2.0f + nj) / ((float)L + 2.0f + nj);
  float ploop = 1.0f - pmove;
  for (int s : {1, 2, 3}) { xsc[s][0] = (float)std::log((double)ploop); xsc[s][1] = (float)std::log((double)pmove); }
}

// Log-odds match scores of node k over all 29 symbols (modelconfig.c:139-146).
void match_logodds(const bath_hmm &h, int k, float sc[BATH_KP_AMINO]) {
  sc[20] = sc[27] = sc[28] = kNegInf;
  for (int x = 0; x < 20; x++) sc[x] = (float)std::log((double)h.mat[(size_t)k * 20 + x] / kAminoBg[x]);
  fill_degenerate_scores(sc);
}

}  // namespace bath

extern "C" void bath_profile_destroy(bath_profile *gm) {
  if (!gm) return;
  delete[] gm->tsc; delete[] gm->rsc; delete gm;
}

extern "C" int bath_profile_config(const bath_hmm *hmm, int L, bath_profile **ret) {
  const int M = hmm->M, Kp = BATH_KP_AMINO;
  bath_profile *gm = new bath_profile();
  std::memset(gm, 0, sizeof *gm);
  gm->M = M; gm->max_length = hmm->max_length; gm->nj = 1.0f;
  std::memcpy(gm->evparam, hmm->evparam, sizeof gm->evparam);
  std::memcpy(gm->compo, hmm->compo, sizeof gm->compo);
  gm->tsc = new float[(size_t)M * 8];
  core_transitions(*hmm, gm->tsc);
  const size_t row = (size_t)(M + 1) * 2;
  gm->rsc = new float[(size_t)Kp * row];
  std::fill(gm->rsc, gm->rsc + (size_t)Kp * row, kNegInf);
  float sc[BATH_KP_AMINO];
  for (int k = 1; k <= M; k++) {
    match_logodds(*hmm, k, sc);
    for (int x = 0; x < Kp; x++) gm->rsc[x * row + 2 * k] = sc[x];
  }
  // inserts score 0 except I_M and the non-residue symbols (modelconfig.c:162-169)
  for (int x = 0; x < Kp; x++) {
    bool residue = (x != 20 && x != 27 && x != 28);
    for (int k = 1; k < M; k++) gm->rsc[x * row + 2 * k + 1] = residue ? 0.0f : kNegInf;
  }
  gm->xsc[0][0] = gm->xsc[0][1] = (float)-0.69314718055994529;   // E loop/move, multihit
  length_model(gm->xsc, gm->nj, L);
  gm->L = L;
  *ret = gm;
  return BATH_OK;
}

extern "C" void bath_fs_profile_destroy(bath_fs_profile *gm) {
  if (!gm) return;
  delete[] gm->tsc; delete[] gm->rsc; delete[] gm->codons; delete[] gm->indel_pos; delete gm;
}

extern "C" int bath_fs_profile_config(const bath_hmm *hmm, const uint8_t basic[64], int codon_lengths, int L_amino,
                                      bath_fs_profile **ret) {
  if (codon_lengths != 3 && codon_lengths != 5) return BATH_EINVAL;
  const int M = hmm->M, Kp = BATH_KP_AMINO, STOP = 27, XAA = 26;
  const bool five = (codon_lengths == 5);
  const int NC = five ? 1367 : 338;
  const size_t W = (size_t)M + 1;

  bath_fs_profile *gm = new bath_fs_profile();
  std::memset(gm, 0, sizeof *gm);
  gm->M = M; gm->max_length = hmm->max_length; gm->codon_lengths = codon_lengths; gm->maxcodons = NC;
  gm->nj = 1.0f; gm->fsprob = hmm->fsprob;
  std::memcpy(gm->evparam, hmm->evparam, sizeof gm->evparam);
  std::memcpy(gm->compo, hmm->compo, sizeof gm->compo);
  gm->tsc = new float[(size_t)M * 8];
  core_transitions(*hmm, gm->tsc);
  gm->rsc = new float[(size_t)(NC + Kp) * W];
  std::fill(gm->rsc, gm->rsc + (size_t)(NC + Kp) * W, kNegInf);
  gm->codons = new uint8_t[(size_t)NC * W]();
  gm->indel_pos = new uint8_t[(size_t)NC * W]();

  // penalties, modelconfig.c:243-254
  const float one_indel = (float)std::log((double)hmm->fsprob);
  const float stop_pen  = (float)std::log((double)hmm->fsprob);
  const float two_indel = five ? (float)std::log(hmm->fsprob / 2.) : 0.f;
  const float no_indel  = five ? (float)std::log(1. - hmm->fsprob * 4.) : (float)std::log(1. - hmm->fsprob * 3.);

  // quasi-codon row indices, hmmer.h:292-316
  auto q1 = [](int x) { return x * 341; };
  auto q2 = [&](int w, int x) { return five ? x * 341 + w * 85 + 1 : x * 84 + w * 21; };
  auto q3 = [&](int v, int w, int x) { return five ? x * 341 + w * 85 + v * 21 + 2 : x * 84 + w * 21 + v * 5 + 1; };
  auto q4 = [&](int u, int v, int w, int x) { return five ? x * 341 + w * 85 + v * 21 + u * 5 + 3 : x * 84 + w * 21 + v * 5 + u + 2; };
  auto q5 = [](int t, int u, int v, int w, int x) { return x * 341 + w * 85 + v * 21 + u * 5 + t + 4; };
  enum { L___X, L_X__, L_XX_, L_X_X, L__XX, L_XXX, L_XXx, L_XxX, L_xXX, L_xxx, L_XXxX, L_XxXX, L_xXXX, L_XXxxX, L_XxxXX, L_xxXXX };

  float sc[BATH_KP_AMINO];
  for (int k = 1; k <= M; k++) {
    match_logodds(*hmm, k, sc);
    for (int x = 0; x < Kp; x++) gm->rsc[(size_t)(NC + x) * W + k] = sc[x];
    auto cell = [&](int c) -> float & { return gm->rsc[(size_t)c * W + k]; };
    auto improve = [&](int c, int aa, int label) {        // keep the best-scoring consistent amino acid
      if (sc[aa] > cell(c)) { cell(c) = sc[aa]; gm->codons[(size_t)k * NC + c] = (uint8_t)aa; gm->indel_pos[(size_t)k * NC + c] = (uint8_t)label; }
    };
    auto assign = [&](int c, int aa, int label, float pen) {
      cell(c) = sc[aa] + pen; gm->codons[(size_t)k * NC + c] = (uint8_t)aa; gm->indel_pos[(size_t)k * NC + c] = (uint8_t)label;
    };
    for (int x = 0; x < 4; x++) for (int w = 0; w < 4; w++) for (int v = 0; v < 4; v++) {
      const int aa = basic[16 * v + 4 * w + x];
      if (five) { improve(q1(x), aa, L___X); improve(q1(v), aa, L_X__); }
      improve(q2(w, x), aa, L__XX);
      improve(q2(v, x), aa, L_X_X);
      improve(q2(v, w), aa, L_XX_);
      const int c3 = q3(v, w, x);
      if (aa == STOP) {                                   // stop codon: best single substitution
        for (int s = 0; s < 4; s++) {
          improve(c3, basic[16 * s + 4 * w + x], L_xXX);
          improve(c3, basic[16 * v + 4 * s + x], L_XxX);
          improve(c3, basic[16 * v + 4 * w + s], L_XXx);
        }
      } else assign(c3, aa, L_XXX, 0.f);
      for (int u = 0; u < 4; u++) {
        const int c4 = q4(u, v, w, x);
        improve(c4, basic[16 * u + 4 * v + x], L_XXxX);
        improve(c4, basic[16 * u + 4 * w + x], L_XxXX);
        improve(c4, basic[16 * v + 4 * w + x], L_xXXX);
        if (five)
          for (int t = 0; t < 4; t++) {
            const int c5 = q5(t, u, v, w, x);
            improve(c5, basic[16 * t + 4 * u + x], L_XXxxX);
            improve(c5, basic[16 * t + 4 * w + x], L_XxxXX);
            improve(c5, basic[16 * v + 4 * w + x], L_xxXXX);
          }
      }
    }
    for (int x = 0; x < 4; x++) {                         // indel / stop penalties
      if (five) cell(q1(x)) += two_indel;
      for (int w = 0; w < 4; w++) {
        cell(q2(w, x)) += one_indel;
        for (int v = 0; v < 4; v++) {
          cell(q3(v, w, x)) += (basic[16 * v + 4 * w + x] == STOP) ? stop_pen : no_indel;
          for (int u = 0; u < 4; u++) {
            cell(q4(u, v, w, x)) += one_indel;
            if (five) for (int t = 0; t < 4; t++) cell(q5(t, u, v, w, x)) += two_indel;
          }
        }
      }
    }
    if (five) { assign(1364, XAA, L_xxx, no_indel); assign(1365, XAA, L_xxx, one_indel); assign(1366, XAA, L_xxx, two_indel); }
    else      { assign(336,  XAA, L_xxx, no_indel); assign(337,  XAA, L_xxx, one_indel); }
  }
  gm->xsc[0][0] = gm->xsc[0][1] = (float)-0.69314718055994529;
  length_model(gm->xsc, gm->nj, L_amino);
  gm->L = L_amino;
  *ret = gm;
  return BATH_OK;
}
