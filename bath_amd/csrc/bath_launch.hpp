// bath_launch.hpp -- host-side launchers shared between bath_filters.hip and bath_pipeline.hip.
#pragma once
#include <functional>
#include <utility>

#include "bath_common.hpp"

// <NR, G> tile shapes of the SSV kernels: NR packed registers per lane, G lanes per target
#define BATH_SSV_SHAPES(X)                                                                                   \
  X(16, 1) X(20, 1) X(24, 1) X(28, 1) X(32, 1) X(36, 1) X(40, 1) X(44, 1) X(48, 1) X(52, 1) X(56, 1) X(60, 1) X(64, 1)     \
  X(68, 1) X(72, 1) X(76, 1) X(80, 1) X(84, 1) X(88, 1) X(92, 1) X(96, 1) X(100, 1) X(104, 1) X(108, 1) X(112, 1)           \
  X(128, 1) X(144, 1) X(160, 1) X(176, 1) X(192, 1) X(208, 1)                                                \
  X(40, 2) X(48, 2) X(56, 2) X(64, 2) X(72, 2) X(76, 2)                                                     \
  X(112, 2) X(128, 2) X(144, 2) X(160, 2) X(176, 2) X(192, 2) X(208, 2)                                      \
  X(112, 4) X(128, 4) X(144, 4) X(160, 4) X(176, 4) X(192, 4) X(208, 4)                                      \
  X(112, 8) X(128, 8) X(144, 8) X(160, 8) X(176, 8) X(192, 8) X(208, 8)

namespace bath {

struct Cand {
  int64_t *window;  int32_t *sf;      // strand*3+frame
  int32_t *startj;  int32_t *len;     // first codon index within the stream, length in aa
  int16_t *v;                         // raw SSV maximum
  int64_t *off;                       // offset of the amino-acid sequence in the pool
  int32_t *msv_status, *vit_status, *fwd_status, *stage, *flags;
  float *usc, *nullsc, *filtersc, *vfsc, *fwdsc;
  double *P;
  int32_t *kminmax;                   // [2*cap]
};

// ORF records of a cascade pass, built and ordered on the device (bath_records.hip): one bath_orf_result per candidate with
// stage >= 1, ordered by (window, strand, frame, start); *d_out stays valid until the next call on ctx
int build_orf_records(bath_hip_ctx *ctx, const Cand &cand, int nc, int64_t window_offset, bath_orf_result **d_out, int64_t *n_out);

struct VitWindowArgs {            // p7_ViterbiFilter_BATH's extra inputs/outputs (vitfilter.c:286)
  double invP_vit, invP_msv;
  const float *d_filtersc;        // [n] indexed by sequence id
  const uint8_t *d_ssv_scores;    // [(M+1)*Kp]
  void *d_wins;                   // WindowRec[win_cap]
  int *d_win_count;
  int win_cap;
  int32_t *d_kminmax;             // [2n] min start node / max end node over the target's windows
};

int launch_ssv_lane(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, const int32_t *d_order, int16_t *d_v);
int launch_ssv_classify(bath_hip_ctx *ctx, const bath_hip_oprofile *om, int64_t n, const int32_t *d_len, const int16_t *d_v, float *d_sc, int32_t *d_status);
int launch_msv_wave(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, const int32_t *d_todo, int64_t ntodo, float *d_sc, int32_t *d_status, const int *ntodo_dev, bool lane_ok = true);
// the same with a lane per target (bath_msv_lane.hip): BATH_OK, an error, or BATH_ENORESULT when the model does not fit a lane's tile
int launch_msv_lane(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, const int32_t *d_todo, int64_t ntodo, float *d_sc, int32_t *d_status, const int *ntodo_dev);
int launch_vit_wave(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, const int32_t *d_todo, int64_t ntodo, float *d_sc, int32_t *d_status,
                    const VitWindowArgs *wa, const int *ntodo_dev);
int launch_fwd_wave(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, const int32_t *d_todo, int64_t ntodo, float *d_sc, int32_t *d_status, const int *ntodo_dev,
                    float *d_xmx = nullptr, const int64_t *d_xmx_off = nullptr, float *d_dp = nullptr, const int64_t *d_dp_off = nullptr, int unihit = 0,
                    const int32_t *d_cfg_len = nullptr);
int launch_bwd_wave(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, int64_t n, const float *d_fwd_xmx, const int64_t *d_xmx_off,
                    float *d_sc, int32_t *d_status, float *d_bck_xmx, float *d_dp = nullptr, const int64_t *d_dp_off = nullptr, int unihit = 0);
int launch_bias_lane(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, const float *d_eo, int eo_stride, const int32_t *d_todo, int64_t ntodo,
                     float *d_nullsc, float *d_filtersc);

int vit_lane_supported(const bath_hip_oprofile *om);
int launch_len_sort(bath_hip_ctx *ctx, const int32_t *d_todo, const int *d_ntodo, const int32_t *d_len, int *d_bins, int32_t *d_sorted);
int launch_vit_lane(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, const int32_t *d_todo, int64_t ntodo, const int *ntodo_dev,
                    float *d_sc, int32_t *d_status, const VitWindowArgs *wa, const int *skip_dev = nullptr);
const int *len_sort_count_longer(const int *d_bins, int T);     // device pointer: after launch_len_sort, the number of targets longer than T
struct MsvConsts;
MsvConsts msv_consts(const bath_hip_oprofile *om);

// ---- the cascade, returning the ORFs that pass the Forward filter (bath_pipeline.hip); *d_pool stays valid until the next pipeline call on ctx
int pipeline_filters_survivors(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *dna, const bath_pipeline_params *prm,
                               bath_pipeline_stats *stats, std::vector<PipelineSurvivor> *out, const uint8_t **d_pool);

// ---- DNA regions gathered into a pool for the frameshift kernels (bath_pipeline.hip)
struct FsWinDev {                  // one DNA window / envelope, device view
  int64_t src_off;                 // offset of its sequence in the DNA block
  int64_t dst_off;                 // offset of the region's copy in the pool
  int32_t seq_n, start, len, strand, kmin, kmax;   // start: 1-based on the strand being read
};
int fs_gather_view(bath_hip_ctx *ctx, const bath_hip_seqs *dna, std::vector<FsWinDev> &regs, const uint8_t *d_comp, bath_hip_seqs *view, const FsWinDev **d_desc_out);

// ---- the DNA windows of the frameshift stage built on the device (bath_fs_windows.hip)
struct WindowRec;                  // bath_kernels.hpp
struct FsCandRec { int64_t window, aa_off; double P; int32_t cand, sf, startj, len; float fwdsc, nullsc; int64_t fxoff; };   // an ORF that passed F4, as a cascade lane leaves it
struct FsLaneSurv {                // one cascade lane's F4 survivors and their hit windows, in the lane's device memory
  const FsCandRec *d_c; const WindowRec *d_w;
  int32_t n_c, n_w, cand_base;     // counts; what makes the lane's candidate ids the block's
  int64_t first_window, dpool;     // ... its sequence indices the block's, its pool offsets relative to the first lane's pool
};
struct FsOrfDev {                  // an ORF that passed F4, in the order esl_gencode emits them per (sequence, strand)
  int64_t w, aa_off; double P;
  int32_t cand, strand, start, end, n; float fwd_null;
  int32_t wb, we;                  // its hit windows, a range of the ordered hit-window list
  int32_t dw_n, dw_len, dw_k;      // the DNA window it asks for (start on the strand, length; node of its best hit window)
  int32_t kmin, kmax;              // first / last model node over its hit windows
  int32_t g0, g1;                  // its (sequence, strand) group, a range of the ordered ORF list
  int32_t pad_;
};
struct FsWinBuild {                // fs_build_windows_device's result: host copies (page-locked, valid until the next build) and device arrays
  int32_t n_orfs = 0, nw = 0, maxlen = 0;
  int64_t pool_bytes = 0, total = 0;
  const int64_t *h_voff = nullptr; const int32_t *h_vlen = nullptr;   // [nw] the windows' pool offsets and lengths: valid when the build returns
  const FsOrfDev *h_orfs = nullptr;         // [n_orfs]  } in flight when the build returns: valid after the next synchronize of
  const int32_t *h_grp = nullptr;           // [2 nw]    } the context's stream (fs_decide_device); a window's group = a range of h_orfs
  bath_fs_window *d_out = nullptr;          // the records on the device (fs_decide_device completes them)
  const FsWinDev *d_desc = nullptr;         // the gather kernel's descriptors
  const int64_t *d_voff = nullptr; const int32_t *d_vlen = nullptr;   // the view's off[] / len[]
};
bool fs_windows_on_device();       // false under BATH_HIP_FS_WINDOWS_HOST=1 (A/B: the host path of rounds 1-5)
// BATH_OK; BATH_ENORESULT: an input this path does not take (the caller runs the host path); an error
int fs_build_windows_device(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_fsprofile *om_fs3, const bath_hip_seqs *dna,
                            const bath_pipeline_params *prm, const FsLaneSurv *lanes, int nlanes, int nc_total, FsWinBuild *out);
// after the parsers: scores -> P-values -> branch on the device, the completed records into h_out[nw]
int fs_decide_device(bath_hip_ctx *ctx, const bath_hip_fsprofile *om_fs3, const bath_pipeline_params *prm, const FsWinBuild &B, const float *d_bias,
                     const float *d_fsc, bath_fs_window *h_out);
float flogsum_host(float a, float b);                          // p7_FLogsum with its table, on the host

// ---- frameshift helpers for the pipeline (bath_frameshift.hip)
int fs_fork(bath_hip_ctx *ctx);   // the context's side stream waits for what the main stream holds so far
int fs_join(bath_hip_ctx *ctx);   // ... and the main stream for the side stream
int fs3_forward_scores(bath_hip_ctx *ctx, const bath_hip_fsprofile *om3, const bath_hip_seqs *dna, float *sc,
                       const std::function<int()> *after_launch = nullptr);   // table log-sum, host array out; <after_launch> runs between the launch and the wait
const float *fsprofile_evparam(const bath_hip_fsprofile *om);
int fs_max_regions();
int fs3_regions(bath_hip_ctx *ctx, const bath_hip_fsprofile *om3, const bath_hip_seqs *dna, float loop, int32_t *regions_out, float *fwd_sc_out = nullptr, const int32_t *kept = nullptr);   // parsers + domain decoding + region heuristics (+ the Forward scores)
int fsprofile_codon_lengths(const bath_hip_fsprofile *om);
int fs3_backward_spec(bath_hip_ctx *ctx, const bath_hip_fsprofile *om3, const bath_hip_seqs *dna, int k);   // speculative Backward of the k longest windows, on ctx->spec_stream
struct FsHostTables { int M, max_length, maxcodons; const float *tsc; const uint8_t *codons; const float *evparam; };
const FsHostTables fsprofile_host(const bath_hip_fsprofile *om);
struct FsTraceOut {               // what the pipeline keeps of an envelope's OA trace
  int32_t ihmm, jhmm, iali, jali, nshift, ok; float domcorrection;
  int32_t ncol, exact, nstops;    // alignment display: columns (first to last match state), identities with the consensus, stop codons
  float aliscore;                 // p7_pli_computeAliScores_BATH: sum of the per-column scores; negative = the domain is dropped
  int32_t col_off;                // where this envelope's columns start in the batch's dense column arrays
};
// <cons>: device array [M+1] of consensus residue codes or nullptr; <steps>/<step_off>: per alignment column
// state | codon length << 4 | indel label << 8 (the columns of all envelopes lie densely; those of envelope e start at (*step_off)[e])
int fs5_envelopes_ex(bath_hip_ctx *ctx, const bath_hip_fsprofile *om5, const bath_hip_seqs *dna, int logsum_mode, int c5_compat,
                     bath_fs5_result *res, float *pp, float *oa, float *ppx, float *oax, FsTraceOut *trace,
                     const uint8_t *cons = nullptr, std::vector<uint16_t> *steps = nullptr, std::vector<int64_t> *step_off = nullptr,
                     std::vector<float> *step_pp = nullptr /* tr->pp of every column, parallel to steps */);

// ---- multi-domain regions (bath_ensemble.hip, host): envelopes and per-residue null2 scores from 200 stochastic tracebacks
int region_trace_ensemble(const bath_hip_oprofile *om, int cfg_L, const uint8_t *res, int Lr, const float *fwd, const float *fx,
                          std::vector<float> *n2sc, std::vector<std::pair<int, int>> *env, uint32_t seed = 42);

int fs_region_trace_ensemble(int M, const float *tsc, float xNL, float xNM, float xE, int ireg, int Lr, const float *fwd, const float *fx,
                             std::vector<std::pair<int, int>> *env, uint32_t seed = 42);
int fs5_region_forward(bath_hip_ctx *ctx, const bath_hip_fsprofile *om5, const bath_hip_seqs *dna, int cfg_len_amino,
                       const float **fwd, std::vector<int64_t> *fwd_off, const float **xmx, std::vector<int64_t> *xmx_off, std::vector<float> *sc,
                       const int **done_flags, const float **sc_live);

// ---- six-frame translation + ORF work list (bath_orfs.hip)
struct OrfRec {                   // one ORF of the length-sorted work list
  int64_t aa_off;                 // offset of its first residue in the amino-acid stream pool
  int32_t w;                      // window
  int32_t len_sf;                 // length | (strand*3+frame) << 28
};
__host__ __device__ inline int orf_stream_pitch(int n) { return (n / 3 + 16) & ~15; }   // bytes reserved per frame of an n-nt window
size_t orf_aa_bytes(const bath_hip_seqs *dna);
int orf_slot_cap(int minlen);                                   // ORF records reserved per tile (all six frames)
int orf_tiles_ensure(bath_hip_ctx *ctx, const bath_hip_seqs *dna);   // fills dna->ntiles, d_tile_desc, d_tile_first
struct OrfTablesDev { const uint8_t *full, *fwd, *rev, *comp, *is_init; bool using_initiators; };   // 18^3 general table, canonical 64-entry tables per strand, complement, initiation codons
int orf_tables_upload(bath_hip_ctx *ctx, int ncbi_table, OrfTablesDev *t, int initiator = 0);
struct OrfBuffers {               // device buffers of one translation pass
  uint8_t *aa;                    // orf_aa_bytes()
  void *slots;                    // ntiles*orf_slot_cap() records of 8 bytes
  void *cross;                    // ntiles*6 records of 8 bytes: ORFs crossing tile edges
  int32_t *cnt, *prefix, *suffix; // ntiles*6 each (cnt uses ntiles)
  int *hist, *cursor, *ntotal;    // kOrfBins, kOrfBins, 1
  OrfRec *sorted;                 // the work list, longest ORFs first; *ntotal entries
};
void orf_buffers_carve(OrfBuffers *ob, void *aa, void *slots, void *sorted, void *misc /* (5*nent + 2*kOrfBins + 64) ints */, size_t nent);
int launch_orf_scan(bath_hip_ctx *ctx, const bath_hip_seqs *dna, const OrfTablesDev &tt, int minlen, const OrfBuffers &b,
                    unsigned long long *d_n_orfs, unsigned long long *d_orf_res, int strands = 0);

}  // namespace bath
