/*
 * torchain_hip.h -- C ABI of the MI355X-native LF-MMI chain loss (libtorchain_hip.so).
 *
 * This is the drop-in boundary for ONE path of nttcslab-sp/torchain: the chain objective and its
 * derivative.  Every entry point names the reference interface it replaces (paths relative to the
 * reference repository).  Plain pointers and sizes only: no torch / TH types, no C++ exceptions
 * across the boundary; all functions that can fail return 0 (TC_OK) or a negative TC_ERR_* code.
 *
 * Conventions (same as the reference, torchain/functions.py:27-30, src/my_lib_chain.cpp:104-136):
 *   nnet_output is a (num_sequences*frames_per_sequence) x num_pdfs fp32 matrix in DEVICE memory,
 *   row = frame * num_sequences + sequence ("all sequences for frame 0; all for frame 1; ..."),
 *   unit column stride, row stride >= num_pdfs (in elements).  All tensors are borrowed for the
 *   duration of the call and never retained.  Work is enqueued on `stream` (a hipStream_t passed as
 *   void*) of `device`; no entry point synchronises the host.
 */
#ifndef TORCHAIN_HIP_H_
#define TORCHAIN_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TC_OK 0
#define TC_ERR_INVALID_ARGUMENT (-1)  /* null pointer, bad dims/stride, leaky not in (0,1)        */
#define TC_ERR_BAD_FST (-2)           /* state/label out of range, arcs not state-major, ...      */
#define TC_ERR_UNSUPPORTED (-3)       /* graph or supervision too large for the on-chip layout    */
#define TC_ERR_WORKSPACE (-4)         /* workspace missing or smaller than tc_*_workspace_bytes   */
#define TC_ERR_HIP (-5)               /* a HIP runtime call failed (see tc_last_hip_error)        */
#define TC_ERR_IO (-6)                /* file could not be read / is not an OpenFst vector FST    */
#define TC_ERR_NOT_SEPARABLE (-7)     /* merged supervision FST does not factor per sequence      */

typedef struct tc_den_graph tc_den_graph;     /* replaces the void* to kaldi::chain::DenominatorGraph */
typedef struct tc_supervision tc_supervision; /* replaces the void* to kaldi::chain::Supervision      */

/* Replaces my_lib_test_chain (src/my_lib.h:45, src/my_lib_chain.cpp:138-213), the reference's native self test: runs
 * the full objective on a small synthetic problem on `device` and checks the properties its Kaldi tests assert
 * (src/chain-supervision-test.hpp:239-341, 388-463).  Returns TC_OK, a negative TC_ERR_* if the library could not
 * run, or the number of the property that failed: 1 weight = w * S * T; 2 objf <= 0 for a numerator inside the
 * denominator; 3 derivative rows sum to ~0; 4 finite differences agree with the derivative; 5 the objective is the
 * failure value -10 * weight.  report6 (nullable, host): objf, weight, worst |row sum|, predicted and observed change
 * of objf under the test perturbation, l2_term.  A C-only integrator can validate an install with this one call. */
int tc_self_test(int device, void *stream, float *report6);

const char *tc_strerror(int code);
int tc_version(void);
int tc_last_hip_error(void);

/* ---- denominator graph ------------------------------------------------------------------- */

/* Replaces my_lib_denominator_graph_new (src/my_lib.h:29, src/my_lib_example.cpp:129-134) for an FST
 * already in memory.  Arcs are listed state-major in the order an OpenFst ArcIterator yields them;
 * ilabel = pdf_id + 1; weights are tropical (-log prob); final_weight[s] = +inf for a non-final
 * state.  Builds the transition tables, the 100-iteration initial-probability estimate and the
 * wavefront schedules used by the kernels.  Host only; touches no GPU. */
int tc_den_graph_create(tc_den_graph **out, int32_t num_states, int64_t num_arcs, const int32_t *arc_src,
                        const int32_t *arc_dst, const int32_t *arc_ilabel, const float *arc_weight,
                        const float *final_weight, int32_t start_state, int32_t num_pdfs);

/* Replaces my_lib_denominator_graph_new(rxfilename, num_pdf) (src/my_lib.h:29) for a den.fst on disk:
 * reads an OpenFst binary VectorFst<StdArc> (what fst::ReadFstKaldi accepts) from a plain filename or
 * from a Kaldi-style piped rxfilename ("gunzip -c den.fst.gz |") and calls tc_den_graph_create. */
int tc_den_graph_read(tc_den_graph **out, const char *rxfilename, int32_t num_pdfs);

/* Replaces my_lib_denominator_graph_free (src/my_lib.h:30). */
void tc_den_graph_free(tc_den_graph *graph);

int32_t tc_den_graph_num_states(const tc_den_graph *graph);
int64_t tc_den_graph_num_arcs(const tc_den_graph *graph);
int32_t tc_den_graph_num_pdfs(const tc_den_graph *graph);
/* Copies the num_states initial probabilities ([K] DenominatorGraph::InitialProbs) to host memory. */
int tc_den_graph_initial_probs(const tc_den_graph *graph, float *out_host);

/* Uploads the graph's immutable tables to `device` (idempotent, thread-safe).  The hot calls do this
 * on first use; call it at init to keep allocation out of the first step.  One copy per device:
 * this is what makes a single graph handle usable from every rank/GPU (the reference cannot:
 * example/chime5/parallel_train.py:27-31,47). */
int tc_den_graph_prepare(tc_den_graph *graph, int device);

/* ---- supervision -------------------------------------------------------------------------- */

/* Replaces my_lib_supervision_new (src/my_lib.h:21, src/my_lib_example.cpp:71-76): builds the handle
 * from the five fields of [K] chain::Supervision the path uses (weight, num_sequences,
 * frames_per_sequence, label_dim, fst).  The FST is the merged, epsilon-free, time-sorted acceptor
 * in CSR form: state i owns arcs [arc_begin[i], arc_begin[i+1]); start state 0; ilabel = pdf_id + 1;
 * final_weight[i] = +inf when state i is not final.  The handle splits the FST into its
 * num_sequences independent per-sequence acceptors (TC_ERR_NOT_SEPARABLE if it does not factor). */
int tc_supervision_create(tc_supervision **out, float weight, int32_t num_sequences,
                          int32_t frames_per_sequence, int32_t label_dim, int32_t num_states,
                          const int32_t *arc_begin, const int32_t *arc_ilabel, const float *arc_weight,
                          const int32_t *arc_nextstate, const float *final_weight);

/* Merges the supervision FSTs of a minibatch's examples into the one acceptor tc_supervision_create takes: what the
 * reference reaches natively through kaldi::nnet3::MergeChainExamples -> [K] chain::AppendSupervision
 * (src/my_lib_example_rand.cpp:160).  Host memory only, no GPU needed.  The pieces arrive concatenated: piece k has
 * piece_num_states[k] states and spans piece_num_frames[k] = num_sequences * frames_per_sequence frames; its CSR
 * offsets (num_states + 1 entries, starting at 0) follow those of the pieces before it in `arc_begin`, its arcs and
 * final weights likewise (nextstate local to the piece; +inf = not final).  fst::Concat + RmEpsilon + breadth-first
 * renumbering: a final state f of piece k-1 receives copies of piece k's start arcs with weight w_f + arc weight and
 * stops being final, piece k's start state disappears, states are numbered in time order.  The caller provides the
 * output arrays (cap_states + 1 / cap_arcs entries; TC_ERR_WORKSPACE if too small: sum of the pieces' states, and
 * sum of the arcs + for every boundary (final states before it) x (start arcs behind it), always suffice).
 * TC_ERR_BAD_FST: a piece is not a connected acceptor whose paths all have piece_num_frames[k] arcs. */
int tc_supervision_append(int32_t num_pieces, const int32_t *piece_num_states, const int32_t *piece_num_frames,
                          const int32_t *arc_begin, const int32_t *arc_ilabel, const float *arc_weight,
                          const int32_t *arc_nextstate, const float *final_weight, int64_t cap_states,
                          int64_t cap_arcs, int32_t *out_num_states, int64_t *out_num_arcs, int32_t *out_arc_begin,
                          int32_t *out_ilabel, float *out_weight, int32_t *out_nextstate, float *out_final);

/* ---- chain examples ("egs") on the host: reading and merging a minibatch -------------------------------------- */

/* Replaces the reading half of the reference's random-access minibatch reader (src/my_lib_example_rand.cpp:35-177:
 * RandomAccessNnetChainExampleReader::Value per key + kaldi::nnet3::MergeChainExamples): opens the n scp entries
 * (paths[i], byte offsets[i] of the "\0B" binary marker; offsets NULL or < 0 = start of file), parses the binary
 * <Nnet3ChainEg> objects (NnetIo with FM/DM/CM/CM2/CM3 matrices, NnetChainSupervision with a compact_acceptor FST,
 * <DW>/<DW2> deriv weights) and, for n > 1 or merge_single != 0, merges them: every input stacked example by example
 * with n = the example's position, supervisions appended as by tc_supervision_append, output indexes and
 * deriv_weights frame-major.  Host memory only, no GPU, thread-safe; TC_ERR_IO (file), TC_ERR_BAD_FST (format; text in
 * tc_example_last_error, per thread), TC_ERR_INVALID_ARGUMENT (examples that cannot be merged). */
typedef struct tc_example tc_example;
int tc_example_read(const char *const *paths, const int64_t *offsets, int32_t n, int merge_single, tc_example **out);
void tc_example_free(tc_example *example);
const char *tc_example_last_error(void);
/* Sequential archives, what the reference reads through SequentialNnetChainExampleReader (src/my_lib_example.cpp:35-69):
 * rxfilename is a path or "command |" (popen).  tc_archive_next returns 1 and one example as stored (not merged; its key,
 * NUL-terminated, in key[0..key_cap)), 0 at the end of the archive, or a negative TC_ERR_* (text in
 * tc_example_last_error).  tc_archive_close returns TC_ERR_IO when a piped command exited with an error. */
typedef struct tc_archive tc_archive;
int tc_archive_open(const char *rxfilename, tc_archive **out);
int tc_archive_next(tc_archive *archive, char *key, int32_t key_cap, tc_example **out);
int tc_archive_close(tc_archive *archive);
/* out2 = {number of inputs, number of outputs}. */
int tc_example_counts(const tc_example *example, int32_t *out2);
/* Input j (reference my_lib_example_feats, src/my_lib_example.cpp:79-100): pointers into the object, valid until
 * tc_example_free; indexes = num_indexes x (n, t, x); features = rows x cols, row-major.  Any out pointer may be NULL. */
int tc_example_input(const tc_example *example, int32_t j, const char **name, int32_t *rows, int32_t *cols,
                     int32_t *num_indexes, const float **features, const int32_t **indexes);
/* Output j (reference my_lib_supervision_new / my_lib_example_reader_indexes, src/my_lib_example.cpp:71-76, 102-127):
 * dims5 = {num_sequences, frames_per_sequence, label_dim, num_states, num_arcs}; the FST arrays are what
 * tc_supervision_create takes. */
int tc_example_output(const tc_example *example, int32_t j, const char **name, int32_t *num_indexes,
                      const int32_t **indexes, const float **deriv_weights, float *weight, int32_t *dims5,
                      const int32_t **arc_begin, const int32_t **ilabel, const float **arc_weight,
                      const int32_t **nextstate, const float **final_weight);

/* Random-access minibatch reader: my_lib_example_rand_reader_new / _reset / _num_batch / _num_data / _next / _free,
 * my_lib_example_rand_feats and my_lib_supervision_rand_new of the reference (src/my_lib.h:8-17 over RandReader,
 * src/my_lib_example_rand.cpp:35-177).  `scp_path` holds "key path:offset" lines; `len_file` ("" or NULL: scp_path +
 * ".len"; absent: the lengths are read from the examples) holds "key frames_per_sequence" pairs.  Examples of equal
 * length are grouped into minibatches of `batchsize` (the last of a length may be smaller), shuffled inside and across
 * the lengths by a std::mt19937 seeded with `seed`.  tc_rand_reader_new_ordered's `order`:
 *   TC_RAND_ORDER_SORTED (tc_rand_reader_new): lengths ascending, Fisher-Yates with the draw (engine() * n) >> 32 -- the same on
 *                  every platform;
 *   TC_RAND_ORDER_REFERENCE: the reference's statement on its own library calls (src/my_lib_example_rand.cpp:119-141):
 *                  lengths in the iteration order of a std::unordered_map<size_t, ...> filled in file order, every
 *                  length's keys copied out and passed to std::shuffle, the batches passed to std::shuffle -- the batch
 *                  lists of the reference built against the same standard library (libstdc++), seed for seed and epoch
 *                  for epoch; rank / world sharding applies on top.
 * New here:
 *   rank / world : every rank forms the same shuffled list and takes batches rank, rank + world, ...; all ranks get
 *                  floor(batches / world) of them per epoch (reference antecedent: example/chime5/parallel_train.py:26-75)
 *   lookahead    : that many worker threads read, parse, merge ([K] MergeChainExamples) and build the supervision
 *                  handles of the next batches while the caller uses the current one (0: everything inside _next)
 * tc_rand_reader_next: 1 = moved to the next minibatch, 0 = the epoch is over, negative TC_ERR_* (text in
 * tc_rand_reader_last_error).  The reader starts BEFORE its first minibatch.  tc_rand_reader_example: the current merged
 * minibatch (tc_example_input / _output; owned by the reader, valid until the next _next / _reset / _free);
 * tc_rand_reader_supervision_new: a supervision handle of it that the caller frees (tc_supervision_free);
 * tc_rand_reader_take_example: the current minibatch itself, the caller's to free (tc_example_free) -- for callers
 * that keep its arrays beyond the next _next.
 * tc_rand_reader_batch_keys: the keys of this rank's batch `batch` of the current epoch, space-separated, into buf;
 * returns their number.  Host only. */
typedef struct tc_rand_reader tc_rand_reader;
#define TC_RAND_ORDER_SORTED 0
#define TC_RAND_ORDER_REFERENCE 1
int tc_rand_reader_new(const char *scp_path, int seed, int batchsize, const char *len_file, int rank, int world,
                       int lookahead, tc_rand_reader **out);
int tc_rand_reader_new_ordered(const char *scp_path, int seed, int batchsize, const char *len_file, int rank, int world,
                               int lookahead, int order, tc_rand_reader **out);
/* device >= 0: the look-ahead threads also stage every supervision they build for that GPU (tc_supervision_stage). */
int tc_rand_reader_set_device(tc_rand_reader *reader, int device);
int tc_rand_reader_reset(tc_rand_reader *reader);
int tc_rand_reader_num_batch(const tc_rand_reader *reader);
int tc_rand_reader_num_data(const tc_rand_reader *reader);
int tc_rand_reader_next(tc_rand_reader *reader);
int tc_rand_reader_example(const tc_rand_reader *reader, const tc_example **out);
int tc_rand_reader_supervision_new(tc_rand_reader *reader, tc_supervision **out);
int tc_rand_reader_take_example(tc_rand_reader *reader, tc_example **out);
int tc_rand_reader_batch_keys(const tc_rand_reader *reader, int32_t batch, char *buf, int32_t cap);
void tc_rand_reader_free(tc_rand_reader *reader);
const char *tc_rand_reader_last_error(void);

/* Replaces my_lib_supervision_free (src/my_lib.h:22). */
void tc_supervision_free(tc_supervision *supervision);
/* Replace my_lib_supervision_num_pdf / _num_sequence / _num_frame (src/my_lib.h:23-25). */
int32_t tc_supervision_num_pdf(const tc_supervision *supervision);
int32_t tc_supervision_num_sequence(const tc_supervision *supervision);
int32_t tc_supervision_num_frame(const tc_supervision *supervision);
float tc_supervision_weight(const tc_supervision *supervision);

/* Copies the supervision's arc tables to `device` on `stream` (asynchronously: pinned staging; idempotent -- a
 * later call on another stream makes that stream wait for the copy).  The tables live in a slot of a per-device
 * pool: tc_supervision_free returns the slot, and the pool reuses it once the last launch that read it has
 * finished (event query, no synchronisation), so fresh supervisions every minibatch cost no hipMalloc / hipFree
 * after warm-up. */
int tc_supervision_prepare(tc_supervision *supervision, int device, void *stream);
/* The host half of that upload alone -- the tables into the pinned staging of a slot of `device`'s pool -- so that a
 * reader's look-ahead thread can do it ahead of the training thread (tc_rand_reader_set_device); no stream is touched. */
int tc_supervision_stage(tc_supervision *supervision, int device);

/* ---- the hot path -------------------------------------------------------------------------- */

/* Bytes of device scratch the calls below need for this problem size (alpha history etc.; for batches of at most
 * 128 sequences of graphs on the on-chip tied kernel also a second history of the same size: such batches run the
 * forward and the backward recursion of a sequence on two CUs at once -- inside the call, forking to a per-device
 * side stream and joining `stream` again, which HIP-graph capture follows).
 * Sizes, with S sequences, T frames, Hs = graph states rounded up (tied on-chip graphs: to whole planes of 4096):
 *   on-chip graphs          4 (T + 1) S Hs                     the alpha history  (1.27 GB at 256 x 150 x 8192)
 *     tied, Hs <= 8192      + 4 S Hs + ~80 (T + 2) S bytes      one more row, the two-sequence form's normalisers
 *     tied, S <= 128        + 4 (T + 1) S Hs                    the second history of the two-CU form
 *   graphs beyond LDS       4 (T + 1) S' H + 12 S' H + 8 S' P   S' = S rounded up to whole slabs of 16 sequences (32 from
 *                                                              28000 states on): the streamed path's [slab][state][G] matrices
 *                           + 4 T S' P if that is <= 1 GB       exp(y) of every frame, transposed once (else per frame)
 *   tied, 28673..40960 positions  + 8 S (2 Hs + ...)            second half of the gather source, parked row sums (per workgroup)
 * plus a few KB of per-sequence scalars and 4 S (T + 2) bytes of frame sums (used by long utterances only). */
int64_t tc_chain_workspace_bytes(const tc_den_graph *graph, int32_t num_sequences, int32_t frames_per_sequence);

/* Replaces my_lib_ComputeChainObjfAndDeriv (src/my_lib.h:33-42, src/my_lib_chain.cpp:104-136), i.e.
 * [K] chain::ComputeChainObjfAndDeriv:
 *   objf = w*num - w*den;  weight = w*S*T;  l2_term = -0.5*w*l2*sum(y^2)
 *   deriv = w*gamma_num - w*gamma_den - w*l2*y      (written, not accumulated; may be NULL)
 *   xent_deriv = w*gamma_num                        (written when non-NULL)
 * and on NaN/inf or a failed alpha.beta check: deriv = xent_deriv = 0, objf = -10*weight.
 * results_dev3 is DEVICE memory for {objf, l2_term, weight} (the reference writes a CPU
 * THFloatTensor, src/my_lib_chain.cpp:126,130; the host wrapper copies the 12 bytes).
 * Stream capture: this call, tc_chain_objf_and_grad, tc_chain_step and tc_den_forward_backward may be captured in a HIP
 * graph (tested: tests/test_gpu_step.py, tests/test_gpu_tied.py).  Warm the call up once outside the capture (per-device
 * tables, pools and side streams are created on first use).  Inside a capture the supervision's upload becomes a node of
 * the graph -- every replay copies its tables again from the library's pinned block -- and none of the library's events
 * is waited for or recorded on the capturing stream except the fork / join pair of its own side streams; the caller keeps
 * graph, supervision and workspace alive for as long as the graph may be replayed.  (A supervision is new with every
 * minibatch, so a captured TRAINING step is a benchmark's tool; DESIGN.md 7 has what replay saves: host time only.) */
int tc_chain_objf_and_deriv(tc_den_graph *graph, tc_supervision *supervision, const float *nnet_output,
                            int64_t num_rows, int32_t num_cols, int64_t row_stride, float *results_dev3,
                            float *nnet_output_deriv, int64_t deriv_stride, float *xent_output_deriv,
                            int64_t xent_stride, float l2_regularize, float leaky_hmm_coefficient,
                            float xent_regularize, void *workspace, int64_t workspace_bytes, int device,
                            void *stream);

/* The same computation with the outputs in the form the reference's autograd backward returns them
 * (torchain/functions.py:106-115: -mmi_grad for the input, -xent_regularize * xent_grad for xent_input):
 *   grad      = -(w*gamma_num - w*gamma_den - w*l2*y)        xent_grad = -xent_regularize * w*gamma_num
 * so that the wrapper needs no further pass over the matrices (the reference makes one to scale xent_grad and one
 * each to negate).  grad is the exact negative of tc_chain_objf_and_deriv's deriv; results_dev3 is unchanged. */
int tc_chain_objf_and_grad(tc_den_graph *graph, tc_supervision *supervision, const float *nnet_output,
                           int64_t num_rows, int32_t num_cols, int64_t row_stride, float *results_dev3,
                           float *nnet_output_grad, int64_t grad_stride, float *xent_output_grad,
                           int64_t xent_stride, float l2_regularize, float leaky_hmm_coefficient,
                           float xent_regularize, void *workspace, int64_t workspace_bytes, int device,
                           void *stream);

/* ONE call per training step: everything torchain/functions.py:62-115 does around my_lib_ComputeChainObjfAndDeriv --
 * the (B, C, T) -> (T*B, C) copy of `to2d` (functions.py:118-125), the objective, the reference's second call on
 * xent_input when kaldi_way == 0 (functions.py:96-103), the cross-entropy objective, the loss value -objf / weight
 * (functions.py:104) and the matrices in the form and layout backward() returns them (functions.py:106-115).
 *   three_d != 0 : input / xent_input / grad / xent_grad are contiguous (B, C, T) tensors, B = num_sequences of the
 *                  supervision, C = its label_dim, T = its frames_per_sequence;  else 2-D (T*B, C), input and xent_input
 *                  with `row_stride`, grad and xent_grad contiguous.
 *   grad          = -(derivative of the objective w.r.t. input);  xent_grad = -xent_regularize * xent_deriv
 *                   (xent branch: xent_input != NULL and xent_regularize != 0, as in the reference)
 *   results_dev3  : device float[3] {objf, l2_term, weight};  loss_dev1 (nullable): device float[1] = -objf / weight
 *   xent_objf_dev : nullable device double[1] = -xent_regularize * sum(xent_input * xent_deriv) (the sum over xent_grad's
 *                   entries as they are written; the caller divides).  xent_deriv has entries only where the numerator
 *                   has posteriors: a (B, C, T) xent_grad is cleared and those entries written in place, xent_input read
 *                   in place -- no frame-major copies of the two.  kaldi_way == 0: the reference's second call overwrites
 *                   all the first wrote, so only it is made (on xent_input); the objective here is still that of the
 *                   first call's xent_deriv (the numerator alone on `input`)
 *   grad == NULL  : an EVALUATION step (xent_grad must be NULL too): [K] ComputeChainObjfAndDeriv with
 *                   nnet_output_deriv == NULL -- the two forward recursions, results, loss and (when asked for) the
 *                   cross-entropy objective; no backward recursion of the denominator, hence no alpha-beta check (the
 *                   NaN / inf guard on objf stays), nothing written but the scalars.  The reference's validation loop
 *                   (example/chime5/train.py:150-171, under torch.no_grad()) pays for a training step because
 *                   torchain/functions.py:74,82 fill mmi_grad whatever autograd needs.
 * Workspace: tc_chain_step_workspace_bytes(graph, B, T, three_d, xent branch).  No host synchronisation. */
int64_t tc_chain_step_workspace_bytes(const tc_den_graph *graph, int32_t num_sequences, int32_t frames_per_sequence,
                                      int three_d, int with_xent);
int tc_chain_step(tc_den_graph *graph, tc_supervision *supervision, const float *input, const float *xent_input,
                  int three_d, int64_t row_stride, float l2_regularize, float leaky_hmm_coefficient,
                  float xent_regularize, int kaldi_way, float *grad, float *xent_grad, float *results_dev3,
                  float *loss_dev1, double *xent_objf_dev, void *workspace, int64_t workspace_bytes, int device,
                  void *stream);

/* The benchmarked unit: [K] DenominatorComputation::Forward() + Backward(deriv_weight, deriv)
 * (direct use in the reference: src/chain-supervision-test.hpp:403-414).
 *   logprob_dev   : device double[1], sum over sequences of the denominator log-prob
 *   deriv         : nullable; accumulate != 0 -> deriv += deriv_weight*gamma (Kaldi's semantics),
 *                   accumulate == 0 -> deriv  = deriv_weight*gamma - l2_scale*y (one store per element)
 *   status_dev    : nullable device int32[1], set to 0 when the t=0 checks pass, else nonzero
 */
int tc_den_forward_backward(tc_den_graph *graph, int32_t num_sequences, const float *nnet_output,
                            int64_t num_rows, int32_t num_cols, int64_t row_stride,
                            float leaky_hmm_coefficient, float deriv_weight, float l2_scale, int accumulate,
                            float *deriv, int64_t deriv_stride, double *logprob_dev, int32_t *status_dev,
                            void *workspace, int64_t workspace_bytes, int device, void *stream);

/* [K] NumeratorComputation::Forward() + Backward() (src/chain-supervision-test.hpp:99-107):
 *   logprob_dev : device double[1] = weight * log Z_num
 *   deriv       : nullable; deriv[row, pdf] += weight * occupation at the supervision's (row, pdf)s */
int tc_num_forward_backward(tc_supervision *supervision, const float *nnet_output, int64_t num_rows,
                            int32_t num_cols, int64_t row_stride, float *deriv, int64_t deriv_stride,
                            double *logprob_dev, void *workspace, int64_t workspace_bytes, int device,
                            void *stream);

/* The cross-entropy objective Kaldi's chain trainer reports next to the chain objective ([K]
 * nnet-chain-training.cc: TraceMatMat(xent_output, xent_deriv, kTrans), xent_deriv = w * numerator posteriors);
 * the reference leaves it as a TODO (torchain/functions.py:88-89).  objf_dev: device double[1].  Needs
 * tc_chain_workspace_bytes-sized or at least 4096 bytes of 16-byte aligned device scratch. */
int tc_xent_objf(const float *xent_output, int64_t num_rows, int32_t num_cols, int64_t output_stride,
                 const float *xent_output_deriv, int64_t deriv_stride, double *objf_dev, void *workspace,
                 int64_t workspace_bytes, int device, void *stream);

/* Diagnostics for bench.py / DESIGN.md: copies a few schedule statistics of the graph
 * (out[0]=padded forward arc slots, out[1]=padded backward arc slots, out[2]=LDS bytes of the fused
 * kernel for this graph, out[3]=threads per workgroup, out[4]=forward rows, out[5]=backward rows,
 * out[6], out[7] = forward / backward LDS bank-conflict factor of the placed arc gathers x 1000, where
 * 1000 means conflict-free; out[8] = 1 when the graph is "tied" -- all non-self-loop arcs entering a
 * state carry one pdf, possibly after state splitting -- and runs a factorised on-chip kernel (one-stream forms up to
 * 16384 layout positions, the plane-wise form up to 28672, with its gather source in LDS a half at a time up to 40960), 2 when
 * the graph is too large for the on-chip layouts and runs the streamed kernel, 0 for a general on-chip kernel (up to 8192 states: on
 * owner-computes schedules). */
int tc_den_graph_stats(const tc_den_graph *graph, int64_t *out9);

/* Kernel choice of batches above one sequence per two CUs.  Tied on-chip graphs of up to 8192 positions have two
 * kernels: the fused one (one sequence per workgroup) and the two-sequence one (a pair of sequences on two CUs that
 * share the arc walk), which is the faster from about 11 arcs per state on.  tc_den_graph_prepare times both once per
 * graph and device on a zero-filled scratch batch (one sequence per CU, 48 frames; the scratch, about
 * 2 * CUs * 48 * num_pdfs * 4 bytes plus that batch's workspace, is freed again) and keeps the two-sequence kernel
 * when it is at least 3% faster; tc_debug_set("no_tune", 1) or ("no_pair", 1) before the graph's first use on the
 * device keeps the fused kernel, ("force_pair", 1) selects the other wherever it fits.  Both kernels agree with the
 * reference to the same tolerance; their results differ from each other in the last bits.
 * Reports the choice (1 = two-sequence kernel) and the two times in ms; prepares the graph if it was not. */
int tc_den_graph_tuning(tc_den_graph *graph, int device, int32_t *two_sequence_kernel, float *fused_ms,
                        float *two_sequence_ms);

/* Fixes that choice instead of timing it: two_sequence_kernel = 0 / 1, or -1 to forget a fixed choice (the next
 * tc_den_graph_prepare on the device times again).  Called BEFORE the graph first reaches `device` no timing launch is
 * made at all; called later the choice applies from the next launch on.  This is what makes the choice reproducible:
 * each kernel is bitwise reproducible by itself, the two differ in the last bits, and a timing race decides by clock
 * noise for a graph near the 3 % threshold -- from run to run and from rank to rank.  Since round 5 the LIBRARY keeps the
 * cache of measured choices (csrc/tuning_cache.cpp): tc_den_graph_prepare looks the graph up -- key: tc_den_graph_hash +
 * the device's name -- in the JSON file $TORCHAIN_TUNING_CACHE (else ~/.cache/torchain_amd/tuning.json) before it times
 * anything and stores what it timed, so a caller of this ABI gets the same kernel run after run without calling this
 * function; the Python wrapper applies rank 0's choice on every rank of a data-parallel job
 * (parallel.sync_den_graph_variant).
 * (The reference's call is deterministic for fixed inputs: src/my_lib_chain.cpp:129-131.) */
int tc_den_graph_set_variant(tc_den_graph *graph, int device, int32_t two_sequence_kernel);
/* 64-bit FNV-1a hash of the graph (sizes, arcs, pdfs, probabilities): the key of such a cache.  0 for a null handle. */
uint64_t tc_den_graph_hash(const tc_den_graph *graph);
/* That cache, directly (host only; e.g. to ship the choices measured on one machine to a fleet): 1 / 0 = found / not. */
int tc_tuning_cache_get(uint64_t graph_hash, const char *device_name, int32_t *two_sequence_kernel);
int tc_tuning_cache_put(uint64_t graph_hash, const char *device_name, int32_t two_sequence_kernel, float fused_ms,
                        float two_sequence_ms);

/* ---- layout conversion either side of the path -------------------------- */

/* Replaces `to2d` (torchain/functions.py:118-125: x.permute(2,0,1).contiguous().view(-1, C)):
 * in_bct is a contiguous (B, C, T) device tensor, out2d[(t*B + b) * out_stride + c] = in_bct[b][c][t]. */
int tc_to2d(const float *in_bct, int32_t B, int32_t C, int32_t T, float *out2d, int64_t out_stride, int device,
            void *stream);

/* The way back, fused with the sign / scale the reference applies in separate passes
 * (torchain/functions.py:108-112: -mmi_grad, -xent_regularize * xent_grad, then autograd's inverse permute):
 * out_bct[b][c][t] = scale * in2d[(t*B + b) * in_stride + c],  out_bct contiguous (B, C, T). */
int tc_from2d(const float *in2d, int64_t in_stride, int32_t B, int32_t C, int32_t T, float scale, float *out_bct,
              int device, void *stream);

/* Diagnostic, host only: replays one arc walk from the graph's built schedules exactly as the kernels
 * consume them and returns, per state, direction 0: sum over in-arcs (h->g) of w*gather[h]*pdf_factor[pdf],
 * direction 1: sum over out-arcs (h->g) of w*gather[g]*pdf_factor[pdf]  (gather: num_states floats,
 * pdf_factor: num_pdfs floats, out: num_states floats).  Lets the schedule builder be tested without a GPU;
 * nothing on the hot path calls it. */
/* Diagnostic, host only: process-wide switches; they exist so that tests and profiles can put a graph or a batch on
 * a kernel family it would not normally take.  Read when a graph is BUILT (tc_den_graph_create / _read):
 *   "force_general"  (1: never use the tied-graph kernel)      "force_streamed" (1: alpha/beta in HBM, as for graphs
 *   "no_split"       (1: do not tied-ify nearly tied graphs)                     beyond the on-chip layouts)
 *   "no_pdf_banks", "no_pdf_search", "no_bank_search" (1: skip those placement passes)   "phantom_pdf0" (1: unused positions of a tied layout on pdf 0, as before round 5)
 *     "sched_trace" (1: builder statistics on stderr)
 *   "slab_wide" / "slab_narrow" (1: the streamed path cuts the batch into slabs of 32 / 16 sequences whatever the graph's size)
 *   "old_arrange"    (1: the greedy placement of a half-slot's cells that rounds 1-4 used, instead of round 5's matching:
 *                     HISTORY.md 4.1e)
 *   "no_planes"      (1: tied graphs of 16385..40960 positions take the streamed path, not the plane-wise on-chip kernel)
 *   "no_split_source" (1: tied graphs of 28673..40960 positions take the streamed path at every batch, as before round 6)
 *   "old_general"    (1: general graphs take round 1's on-chip kernel, not the one on owner-computes schedules)
 * Read at launch (one relaxed atomic load):
 *   "no_phase_split" (1: batches of at most 128 sequences of tied on-chip graphs take the fused kernel instead of
 *                     running forward and backward recursion on two CUs at once; the plane-wise kernel likewise)
 *   "no_num_overlap" (1: the numerator always follows the denominator on the caller's stream; by default it runs
 *                     beside it on a side stream when the denominator leaves CUs idle)
 *   "force_pair"     (1: the two-sequence kernel wherever it fits, whatever the batch and the graph's timing said)
 *   "no_mitm" / "force_mitm" (two CUs per sequence: never / always the form that meets in the middle instead of
 *                     two pure recursions and a combining pass; by default from 32 / 48 / 64 sequences by layout class)
 *   "planes_meet_at" (M > 0: the two-workgroup form of the plane-wise kernel meets at frame M instead of T / 2 -- measured:
 *                     T / 2 is the balanced choice, profiles/r06_planes_ab.txt)
 *   "exp_per_frame"  (1: the streamed path transposes exp(y) one frame at a time, as it does when all frames would take
 *                     more than 1 GB of workspace)
 * Read when a graph first reaches a device (tc_den_graph_prepare, see tc_den_graph_tuning):
 *   "no_pair"        (1: never the two-sequence kernel)        "no_tune" (1: no timing launches; the fused kernel)
 * The same switches can be set from the environment when the library is loaded:
 *   TORCHAIN_HIP_DEBUG="no_tune,force_streamed=1"   (unknown keys are reported on stderr and ignored).
 * Returns TC_ERR_INVALID_ARGUMENT for an unknown key. */
int tc_debug_set(const char *key, int value);
/* Diagnostic counters: "pool_device_allocs" = device allocations made so far by the per-device supervision pool
 * (stops growing once the pool is warm: a training step then allocates and frees nothing), "pool_reuses" = slots
 * handed out again; "den_launches" / "den_backward_launches" / "num_launches" / "num_backward_launches" /
 * "layout_launches" = denominator computations enqueued / those with a backward recursion / numerator computations /
 * those with a backward recursion / (B, C, T) <-> (T*B, C) copies, process-wide since the library was loaded;
 * (tc_den_graph_prepare's one-off timing launches of a graph's two kernels are counted like any other);
 * "den_long_utterance_launches" = on-chip denominator launches whose frame sums did not fit LDS and went through the
 * workspace (utterances of many hundred frames on the largest on-chip graphs).
 * -1 for an unknown key. */
int64_t tc_debug_counter(const char *key);

int tc_den_graph_debug_walk(const tc_den_graph *graph, int direction, const float *gather,
                            const float *pdf_factor, float *out);

#ifdef __cplusplus
}
#endif
#endif /* TORCHAIN_HIP_H_ */
