// bath_alidisplay.hip -- the alignment block of a hit, from its trace (host code; nothing here touches the GPU).
//
// Reference: p7_alidisplay_fs_Create (src/p7_alidisplay.c:538-925) and p7_alidisplay_nonfs_Create (:937-1243) build the display
// lines of a domain from dom->tr -- model consensus, match line, translation, the codons with a quasi-codon's missing or extra
// nucleotides marked (nuc_one .. nuc_five, :91-190), posterior-probability digits (p7_alidisplay_EncodePostProb, :3689) -- and
// p7_alidisplay_Print_BATH (:3757-4110) prints them in blocks, five characters per column, with the optional CS / RF lines above
// and the optional frame line (--frameline) below the codons.  This file restates both steps for one trace of
// bath_hip_domain_traces; splice sites (the '$' columns of --splice) are outside this library's path.
// Pinned by every alignment block of the recorded runs (tests/golden/*.out): tests/test_alidisplay_cpu.py (oracle traces),
// tests/test_alidisplay_gpu.py (the GPU path's traces).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "bath_hip.h"

namespace {

const char kAmino[] = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~";
const char kDna[] = "ACGT-RYMKSWHBVDN*~";
enum { I___X = 0, I_X__, I_XX_, I_X_X, I__XX, I_XXX, I_XXx, I_XxX, I_xXX, I_xxx, I_XXxX, I_XxXX, I_xXXX, I_XXxxX, I_XxxXX, I_xxXXX };   // hmmer.h:252-270

void appendf(std::string &out, const char *fmt, ...) {
  char tmp[512];
  va_list ap;
  va_start(ap, fmt);
  const int n = std::vsnprintf(tmp, sizeof tmp, fmt, ap);
  va_end(ap);
  if (n > 0) out.append(tmp, (size_t)std::min<int>(n, (int)sizeof tmp - 1));
}

int integer_textwidth(long n) { int w = (n < 0) ? 1 : 0; while (n != 0) { n /= 10; w++; } return w; }
char encode_pp(float p) { return (p + 0.05 >= 1.0) ? '*' : (char)((char)((p + 0.05) * 10.0) + '0'); }

// get_codon_index, p7_alidisplay.c:32-88
int codon_index(int len, const int *n) {
  bool canon = true;
  for (int q = 0; q < len; q++) canon = canon && n[q] >= 0 && n[q] < 4;
  switch (len) {
    case 1: return canon ? n[0] * 341 : 1366;
    case 2: return canon ? n[1] * 341 + n[0] * 85 + 1 : 1365;
    case 3: return canon ? n[2] * 341 + n[1] * 85 + n[0] * 21 + 2 : 1364;
    case 4: return canon ? n[3] * 341 + n[2] * 85 + n[1] * 21 + n[0] * 5 + 3 : 1365;
    default: return canon ? n[4] * 341 + n[3] * 85 + n[2] * 21 + n[1] * 5 + n[0] + 4 : 1366;
  }
}

char low(char ch) { return (char)std::tolower((unsigned char)ch); }
char sym(int x) { return (x >= 0 && x < 18) ? kDna[x] : '?'; }

// nuc_one .. nuc_five, p7_alidisplay.c:91-190
char nuc_one(int len, int indel, int c1) {
  if (len < 4) return ' ';
  if (indel == I_xXXX || indel == I_xxXXX || indel == I_xxx) return low(sym(c1));
  return sym(c1);
}
char nuc_two(int len, int indel, int c1, int c2) {
  if (len < 4) {
    if (indel == I___X || indel == I__XX) return '-';
    if (indel == I_xXX || indel == I_xxx) return low(sym(c1));
    return sym(c1);
  }
  if (indel == I_XXxX || indel == I_xXXX || indel == I_XXxxX) return sym(c2);
  return low(sym(c2));
}
char nuc_three(int len, int indel, int c1, int c2, int c3) {
  if (len == 1 || indel == I_X_X) return '-';
  if (indel == I__XX) return sym(c1);
  if (len < 4) return (indel == I_XxX || indel == I_xxx) ? low(sym(c2)) : sym(c2);
  if (indel == I_XxXX || indel == I_xXXX || indel == I_xxXXX) return sym(c3);
  return low(sym(c3));
}
char nuc_four(int len, int indel, int c1, int c2, int c3, int c4) {
  if (indel == I___X) return sym(c1);
  if (indel == I_X_X || indel == I__XX) return sym(c2);
  if (len < 3) return '-';
  if (len == 3) return (indel == I_XXx || indel == I_xxx) ? low(sym(c3)) : sym(c3);
  if (indel == I_XXxxX || indel == I_xxx) return low(sym(c4));
  return sym(c4);
}
char nuc_five(int len, int indel, int c5) {
  if (len < 5) return ' ';
  return indel == I_xxx ? low(sym(c5)) : sym(c5);
}

bool dna_has(int x, int y) {
  static const char *members[18] = {"A", "C", "G", "T", "", "AG", "CT", "AC", "GT", "CG", "AT", "ACT", "CGT", "ACG", "AGT", "ACGT", "", ""};
  return x >= 0 && x < 18 && std::strchr(members[x], "ACGT"[y]) != nullptr;
}
// esl_gencode_GetTranslation: the amino acid every expansion of a degenerate codon agrees on, else X
int translate(const uint8_t basic[64], int a, int b, int c) {
  if (a < 4 && b < 4 && c < 4) return basic[16 * a + 4 * b + c];
  int aa = -1;
  for (int x = 0; x < 4; x++) if (dna_has(a, x))
    for (int y = 0; y < 4; y++) if (dna_has(b, y))
      for (int z = 0; z < 4; z++) if (dna_has(c, z)) {
        const int v = basic[16 * x + 4 * y + z];
        if (aa == -1) aa = v; else if (aa != v) return 26;
      }
  return aa == -1 ? 26 : aa;
}

struct Display {                       // P7_ALIDISPLAY's lines, one entry per trace state z1..z2
  std::string model, mline, aseq, ntseq, ppline, csline, rfline;
  std::vector<int> codon;              // ad->codon: the state's codon length; 6 marks a stop codon, 0 a delete state
  int hmmfrom = 0, hmmto = 0;
};

int p7_alidiplay_frame(long nuc_from, long nuc_to) {                    // p7_alidisplay.c:3718
  int frame;
  if (nuc_from < nuc_to) { frame = (int)((nuc_to + 1) % 3); if (frame == 0) frame = 3; }
  else { frame = (int)(-1 * (nuc_to % 3)); if (frame == 0) frame = -3; }
  return frame;
}

}  // namespace

extern "C" int64_t bath_alidisplay_print(const bath_domain_trace *tr, const int8_t *st, const int32_t *k, const int32_t *i, const int8_t *c, const float *pp,
                                         const uint8_t *window_dsq, int32_t window_len, const bath_fs_profile *gm_fs5, const bath_profile *gm,
                                         const uint8_t basic[64], const bath_alidisplay_opts *o, char *buf, int64_t cap) {
  if (!tr || !st || !k || !i || !c || !window_dsq || !o || !o->consensus || tr->N <= 0) return -1;
  if (tr->frameshift ? (!gm_fs5 || gm_fs5->codon_lengths != 5 || !gm_fs5->codons || !gm_fs5->indel_pos) : (!gm || !basic)) return -1;
  const int N = tr->N;
  auto nt = [&](int pos) -> int { return (pos >= 1 && pos <= window_len) ? (int)window_dsq[pos - 1] : 15; };
  Display ad;
  ad.model.assign((size_t)N, ' '); ad.mline.assign((size_t)N, ' '); ad.aseq.assign((size_t)N, ' '); ad.ntseq.assign((size_t)N * 5, ' ');
  ad.codon.assign((size_t)N, 0);
  if (pp) ad.ppline.assign((size_t)N, '.');
  if (o->cs) ad.csline.assign((size_t)N, '.');
  if (o->rf) ad.rfline.assign((size_t)N, '.');
  ad.hmmfrom = k[0]; ad.hmmto = k[N - 1];
  const int M = o->M;
  for (int z = 0; z < N; z++) {
    const int kk = k[z], ii = i[z], s = st[z], cl = c[z];
    if (kk < 0 || kk > M) return -1;
    if (o->cs && s != BATH_T_I) ad.csline[(size_t)z] = o->cs[kk];
    if (o->rf && s != BATH_T_I) ad.rfline[(size_t)z] = o->rf[kk];
    if (pp && s != BATH_T_D) ad.ppline[(size_t)z] = encode_pp(pp[z]);
    char *n5 = &ad.ntseq[(size_t)z * 5];
    if (s == BATH_T_M && tr->frameshift) {                              // p7_alidisplay_fs_Create, :700-760
      if (cl < 1 || cl > 5) return -1;
      int n[5] = {-1, -1, -1, -1, -1};
      for (int q = 0; q < cl; q++) n[q] = nt(ii - cl + 1 + q);
      const int idx = codon_index(cl, n);
      const size_t row = (size_t)kk * (size_t)gm_fs5->maxcodons + (size_t)idx;
      const int aa = gm_fs5->codons[row], indel = gm_fs5->indel_pos[row];
      ad.model[(size_t)z] = o->consensus[kk];
      ad.codon[(size_t)z] = cl;
      n5[0] = nuc_one(cl, indel, n[0]); n5[1] = nuc_two(cl, indel, n[0], n[1]); n5[2] = nuc_three(cl, indel, n[0], n[1], n[2]);
      n5[3] = nuc_four(cl, indel, n[0], n[1], n[2], n[3]); n5[4] = nuc_five(cl, indel, n[4]);
      const char *p = std::strchr(kAmino, std::toupper((unsigned char)o->consensus[kk]));
      const int cons_code = p && *p ? (int)(p - kAmino) : -1;           // esl_abc_DigitizeSymbol (case-insensitive)
      const float msc = gm_fs5->rsc[((size_t)gm_fs5->maxcodons + (size_t)aa) * (size_t)(gm_fs5->M + 1) + (size_t)kk];   // p7P_MSC_AMINO5
      if (aa == cons_code) ad.mline[(size_t)z] = ad.model[(size_t)z];
      else if (expf(msc) > 1.0f) ad.mline[(size_t)z] = '+';
      ad.aseq[(size_t)z] = (char)std::toupper((unsigned char)kAmino[aa]);
      if (cl == 3 && (indel == I_XXx || indel == I_XxX || indel == I_xXX)) ad.codon[(size_t)z] = 6;   // a stop codon
    } else if (s == BATH_T_M) {                                         // p7_alidisplay_nonfs_Create, :1100-1130
      const int a = nt(ii - 2), b = nt(ii - 1), cc = nt(ii);
      int aa = translate(basic, a, b, cc);
      if (o->initiator != BATH_INIT_ANY && tr->win_start + (ii - 2) - 1 == tr->orf_start) aa = 10;   // the ORF's initiation codon reads M (orfsq->dsq)
      ad.model[(size_t)z] = o->consensus[kk];
      ad.codon[(size_t)z] = cl;
      ad.aseq[(size_t)z] = (char)std::toupper((unsigned char)kAmino[aa]);
      n5[1] = (char)std::toupper((unsigned char)sym(a)); n5[2] = (char)std::toupper((unsigned char)sym(b)); n5[3] = (char)std::toupper((unsigned char)sym(cc));
      const char *p = std::strchr(kAmino, std::toupper((unsigned char)o->consensus[kk]));
      const int cons_code = p && *p ? (int)(p - kAmino) : -1;
      const float msc = gm->rsc[(size_t)aa * (size_t)(gm->M + 1) * 2 + (size_t)kk * 2];   // p7_oprofile_FGetEmission = exp of the match log-odds
      if (aa == cons_code) ad.mline[(size_t)z] = ad.model[(size_t)z];
      else if (expf(msc) > 1.0f) ad.mline[(size_t)z] = '+';
    } else if (s == BATH_T_I) {
      const int a = nt(ii - 2), b = nt(ii - 1), cc = nt(ii);
      ad.model[(size_t)z] = '.';
      ad.codon[(size_t)z] = 3;
      if (tr->frameshift) {                                             // :865-885: lower case; a stop codon reads '*'
        const int n[3] = {a, b, cc};
        const size_t row = (size_t)kk * (size_t)gm_fs5->maxcodons + (size_t)codon_index(3, n);
        const int indel = gm_fs5->indel_pos[row];
        int aa = gm_fs5->codons[row];
        if (indel == I_XXx || indel == I_XxX || indel == I_xXX) { ad.codon[(size_t)z] = 6; aa = 27; }
        ad.aseq[(size_t)z] = low(kAmino[aa]);
        n5[1] = sym(a); n5[2] = sym(b); n5[3] = sym(cc);
      } else {                                                          // :1145-1160: upper case
        ad.aseq[(size_t)z] = (char)std::toupper((unsigned char)kAmino[translate(basic, a, b, cc)]);
        n5[1] = (char)std::toupper((unsigned char)sym(a)); n5[2] = (char)std::toupper((unsigned char)sym(b)); n5[3] = (char)std::toupper((unsigned char)sym(cc));
      }
    } else if (s == BATH_T_D) {
      ad.model[(size_t)z] = o->consensus[kk];
      ad.aseq[(size_t)z] = '-';
      n5[1] = n5[2] = n5[3] = '-';
    } else return -1;
  }

  // ---- p7_alidisplay_Print_BATH(fp, ad, 30, 40, textw, pli), without splice sites
  std::string out;
  std::string hmmname = o->hmm_name ? o->hmm_name : "", seqname = o->seq_name ? o->seq_name : "";
  const int max_namewidth = 30, min_aliwidth = 40;
  int namewidth = (int)std::max(hmmname.size(), seqname.size());
  while (namewidth > max_namewidth + 3) {
    std::string &longer = hmmname.size() > seqname.size() ? hmmname : seqname;
    longer = longer.substr(0, (size_t)max_namewidth) + "...";
    namewidth = (int)std::max(hmmname.size(), seqname.size());
  }
  namewidth = std::max(namewidth, 8);
  const int coordwidth = std::max(std::max(integer_textwidth(ad.hmmfrom), integer_textwidth(ad.hmmto)), std::max(integer_textwidth((long)o->sqfrom), integer_textwidth((long)o->sqto)));
  int max_aliwidth = (o->textw > 0) ? o->textw - namewidth - 2 * coordwidth - 5 : N;
  if (max_aliwidth < N && max_aliwidth < min_aliwidth) max_aliwidth = min_aliwidth;
  max_aliwidth -= 4;
  max_aliwidth /= 5;
  if (max_aliwidth < 1) max_aliwidth = 1;
  const bool fwd = o->sqfrom < o->sqto;
  long i1 = (long)o->sqfrom, i2 = fwd ? i1 - 1 : i1 + 1;
  int k1 = ad.hmmfrom;
  std::vector<int> frameline((size_t)max_aliwidth + 1, 0);
  auto annot_line = [&](const std::string &line, int pos, int w, const char *tail) {
    appendf(out, "  %*s ", namewidth + coordwidth + 1, " ");
    out += "  ";
    for (int q = 0; q < w && pos + q < N; q++) { out += "  "; out += line[(size_t)(pos + q)]; out += "  "; }
    out += tail;
  };
  for (int pos = 0; pos < N;) {
    if (pos > 0) out += "\n";
    const int w = max_aliwidth;
    int ni = 0, nk = 0;
    for (int z = pos; z < pos + w && z < N; z++) {
      if (ad.model[(size_t)z] != '.' && ad.model[(size_t)z] != ' ') nk++;
      if (ad.aseq[(size_t)z] != '-') ni++;
    }
    const int k2 = k1 + nk - 1;
    if (!ad.csline.empty()) annot_line(ad.csline, pos, w, "  \n");
    if (!ad.rfline.empty()) annot_line(ad.rfline, pos, w, "   RF\n");
    appendf(out, "  %*s %*d ", namewidth, hmmname.c_str(), coordwidth, k1);
    out += "  ";
    for (int q = 0; q < w && pos + q < N; q++) { out += "  "; out += ad.model[(size_t)(pos + q)]; out += "  "; }
    out += "  ";
    appendf(out, " %-*d\n", coordwidth, k2);
    annot_line(ad.mline, pos, w, "  \n");
    annot_line(ad.aseq, pos, w, "  \n");
    appendf(out, "  %*s", namewidth, seqname.c_str());
    if (ni > 0) appendf(out, " %*ld ", coordwidth, i1); else appendf(out, " %*s ", coordwidth, "-");
    out += "  ";
    for (int j = 0; j < w && pos + j < N; j++) {
      out.append(ad.ntseq, (size_t)(pos + j) * 5, 5);
      const int cd = ad.codon[(size_t)(pos + j)];
      long c1;
      if (fwd) { c1 = i2;     i2 += (cd == 6 ? 3 : cd); }
      else     { c1 = i2 - 1; i2 -= (cd == 6 ? 3 : cd); }
      frameline[(size_t)j] = (cd == 0 || cd == 6) ? 0 : p7_alidiplay_frame(c1, i2);
    }
    out += "  ";
    if (ni > 0) appendf(out, " %-*ld\n", coordwidth, i2); else appendf(out, " %*s\n", coordwidth, "-");
    if (o->show_frameline) {
      appendf(out, "  %*s ", namewidth + coordwidth + 1, "");
      out += "  ";
      for (int j = 0; j < w && pos + j < N; j++) {
        const int f = frameline[(size_t)j];
        if (f > 0) appendf(out, "  %d  ", f);
        else if (f < 0) appendf(out, " %d  ", f);
        else if (ad.codon[(size_t)(pos + j)] == 6) appendf(out, "  %d  ", f);
        else out += "  .  ";
      }
      out += "  ";
      out += " FRAME\n";
    }
    appendf(out, "  %*s ", namewidth + coordwidth + 1, "");
    out += "  ";
    for (int q = 0; q < w && pos + q < N; q++) {
      if (!ad.ppline.empty()) { out += "  "; out += ad.ppline[(size_t)(pos + q)]; out += "  "; }
      else out += "     ";
    }
    out += "  ";
    out += " PP\n";
    k1 += nk;
    i1 = fwd ? i2 + 1 : i2 - 1;
    pos += w;
  }
  if (buf && cap > 0) std::memcpy(buf, out.data(), (size_t)std::min<int64_t>(cap, (int64_t)out.size()));
  return (int64_t)out.size();
}
