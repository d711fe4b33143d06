/*
 * dctr.h -- C-ABI of libdctr_hip.so: the MI355X (gfx950) hot path of DeepCTR-Torch.
 *
 * The reference (shenweichen/DeepCTR-Torch v0.2.9) has no native seam: every op on this path is a
 * chain of ATen calls issued from Python.  This header is the seam a maintainer would bind instead
 * (ctypes stub in INTEGRATION.md).  Each entry point names the reference lines it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (a PyTorch tensor) unless stated
 *     otherwise; the library never allocates, frees or retains memory.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synchronised, so every
 *     call is hipGraph-capturable.
 *   - return value: 0 ok, <0 a DCTR_E* code (bad argument; nothing was launched), >0 the hipError_t
 *     of the failed launch.  Nothing throws.
 *   - all arithmetic is fp32 (the reference computes in fp32).
 *   - ids travel as float32 inside X exactly as in the reference (basemodel.py:242) and are
 *     truncated toward zero like Tensor.long() (basemodel.py:369).
 */
#ifndef DCTR_H
#define DCTR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCTR_ABI_VERSION 25

#define DCTR_OK 0
#define DCTR_EINVAL (-1) /* null / negative / inconsistent argument            */
#define DCTR_ENOSUP (-2) /* shape outside what the kernels are built for        */
#define DCTR_EALIGN (-3) /* pointer or leading dimension violates an alignment  */

typedef void* dctr_stream_t; /* hipStream_t */

/* pooling of a VarLenSparseFeat (inputs.py:41-46, sequence.py:49-77); 0 = plain SparseFeat */
#define DCTR_POOL_NONE 0
#define DCTR_POOL_SUM 1
#define DCTR_POOL_MEAN 2
#define DCTR_POOL_MAX 3

/* One embedding "field": a SparseFeat / VarLenSparseFeat bound to its table.  64 bytes.
 * A deep field gathers dim-wide rows (inputs.py:158-180, basemodel.py:368-375); a wide field is the
 * same thing with dim == 1 (Linear's 1-dim tables, basemodel.py:45-46,65-67).                      */
typedef struct dctr_field {
  float* table;    /* [vocab, dim] rows `ld` floats apart, embedding_dict[embedding_name].weight     */
  float* gacc;     /* [vocab, dim] contiguous gradient slab, zero at rest (NULL when never accumulated) */
  float* state;    /* [vocab, dim] rows `ld_state` floats apart: optimizer state (Adagrad sum) or NULL */
  int64_t vocab;   /* rows                                                                           */
  int32_t dim;     /* embedding_dim (1 for wide)                                                     */
  int32_t col;     /* first column of X holding the id(s)        (inputs.py:99-123)                  */
  int32_t len;     /* 1 for SparseFeat, maxlen for VarLenSparseFeat                                  */
  int32_t pool;    /* DCTR_POOL_*                                                                    */
  int32_t len_col; /* column of X holding the valid length, or -1: mask = (id != 0) (inputs.py:146)  */
  int32_t out_off; /* float offset of this field's slice inside one output row (deep only)          */
  int32_t ld;      /* floats between consecutive rows of `table`; 0 = dim (contiguous).  A table may be a */
  int32_t ld_state;/* strided view of a slab that interleaves a row with its optimizer state (and the    */
                   /* wide weight of the same id): what is updated together then shares one 128-byte line */
} dctr_field_t;

/* ---- general update units (round 5): pooled VarLen fields and shared tables on the deterministic update ------
 * The reference pools a VarLenSparseFeat with SequencePoolingLayer (inputs.py:141-155, sequence.py:49-77) and lets
 * several feature columns share one nn.Embedding through `embedding_name` (inputs.py:158-180); autograd then sends
 * every (sample, position) of every such column to embedding_dense_backward of the ONE table.  Here a UNIT is a
 * (deep table | none, wide table | none) pair with every X column that feeds it -- its SLOTS (virtual columns):
 *   ids_t / parts_t are [n_vcols, B]; a unit's slots are consecutive virtual columns [c0, c0 + n_slots), so its entries
 *   are the contiguous run ids_t[c0 * B ... (c0 + n_slots) * B) and an entry is named by v = slot_local * B + b;
 *   a slot whose position is masked out for a sample (sum / mean pooling: id == 0, or t >= length) carries the tag
 *   0xFFFF in parts_t and is no entry; a max-pooled slot is always an entry, its gradient is masked per element by
 *   the forward's arg-max side output;
 *   a unit of n_slots slots is split over k = n_slots groups of P partitions (rows {id : id mod (k P) == j P + p} belong
 *   to workgroup (vunit (unit, j), p)), so a partition holds ~96 entries whatever the number of slots.
 * A plan without `ext` is the simple case the kernels have always run: unit u = one X column (units[u]) over its own
 * tables, n_vcols = n_vunits = n_units.                                                                              */
typedef struct dctr_uslot {
  int32_t col;      /* X column of the id                                                                            */
  int32_t goff;     /* float offset of the deep field's slice in `out` / `g_out` rows; -1: no deep side              */
  int32_t wide;     /* 1: the entry carries the wide gradient too                                                    */
  int32_t pool;     /* DCTR_POOL_* of the field(s) of this slot                                                      */
  int32_t t;        /* position inside a pooled field (0 otherwise)                                                  */
  int32_t len;      /* positions of the field                                                                        */
  int32_t len_col;  /* X column of the valid length, -1: mask = (id != 0)                                            */
  int32_t den;      /* row of den_t holding mean pooling's divisor count + 1e-8 (sequence.py:72-74); -1              */
  int32_t am_deep;  /* byte offset in an amax row of the deep field's per-element arg-max positions (max pooling); -1 */
  int32_t am_wide;  /* the same for the wide field (one byte); -1                                                    */
  int32_t vu0;      /* first vunit of the slot's unit                                                                */
  int32_t pad_;
} dctr_uslot_t;

typedef struct dctr_vunit {
  int32_t di, wi;        /* field whose descriptor supplies the unit's deep / wide table; -1: none                   */
  int32_t c0, n_slots;   /* the unit's virtual columns                                                               */
  int32_t k, j;          /* the unit has k * P partitions; this vunit's workgroups own [j P, (j + 1) P)              */
  int32_t kshift, pad_;  /* q / k == (q * kmagic) >> kshift for 0 <= q < 2^31 (kshift = 32 + ceil_log2 k)            */
  uint64_t kmagic;
} dctr_vunit_t;

/* host struct; `slots`, `vunits`, `am_deep_off`, `am_wide_off`, `den_t`, `amax` are DEVICE pointers, `h_*` host arrays.
 * den_t / amax are PER-STEP buffers the caller points at before each call (like dctr_plan_t.step_sync):
 *   den_t [n_den, B] float  written by dctr_embed_ids, read by dctr_embed_update
 *   amax  [B, ld_amax] u8   written by dctr_embed_fwd (arg-max position per element of every max-pooled field, first
 *                           maximum wins like torch.max), read by dctr_embed_update                                  */
typedef struct dctr_plan_ext {
  const dctr_uslot_t* slots;    /* [n_vcols]                                                                         */
  const dctr_vunit_t* vunits;   /* [n_vunits]                                                                        */
  const int32_t* am_deep_off;   /* [n_deep] byte offset in an amax row, -1: not max-pooled                           */
  const int32_t* am_wide_off;   /* [n_wide]                                                                          */
  const dctr_vunit_t* h_vunits; /* host copy of vunits                                                               */
  const int64_t* h_vocab;       /* [n_vunits] rows of the vunit's tables                                             */
  float* den_t;
  uint8_t* amax;
  int32_t n_vcols, n_vunits, n_units, max_unit_slots;
  int32_t n_den, ld_amax;
  /* the pooled fields' positions, flattened in field order, for the fused gather + tower launch                     */
  /* (dctr_embed_tower_train_step): entry = (field index << 16) | position                                           */
  const int32_t* gslot_deep;    /* [n_gslot_deep] device                                                             */
  const int32_t* gslot_wide;    /* [n_gslot_wide] device                                                             */
  int32_t n_gslot_deep, n_gslot_wide;
} dctr_plan_ext_t;
#define DCTR_MAX_UNIT_SLOTS 128
size_t dctr_sizeof_uslot(void);
size_t dctr_sizeof_vunit(void);
size_t dctr_sizeof_plan_ext(void);

/* The compiled feature-column schema of one model: what build_input_features + create_embedding_matrix
 * + Linear.__init__ establish in the reference (inputs.py:99-180, basemodel.py:34-61).
 * The struct itself lives in HOST memory; the arrays it points to live on the device.              */
typedef struct dctr_plan {
  const dctr_field_t* deep;   /* [n_deep] fixed-length fields first, then VarLen (basemodel.py:380)  */
  const dctr_field_t* wide;   /* [n_wide]                                                            */
  const int32_t* dense_cols;  /* [n_dense]  X column of every dense scalar of dnn_feature_columns    */
  const int32_t* wdense_cols; /* [n_wdense] X column of every dense scalar of linear_feature_columns */
  const float* wdense_w;      /* [n_wdense] Linear.weight (basemodel.py:58-61), may be NULL          */
  int32_t n_deep;
  int32_t n_deep_fixed;       /* leading deep fields with len == 1                                   */
  int32_t n_wide;
  int32_t n_dense;
  int32_t n_wdense;
  int32_t dense_off;          /* float offset in an output row where the dense block goes, -1: skip  */
  int32_t emb_dim;            /* common dim of all deep fields, 0 if they differ (then no FM)        */
  int32_t n_xcols;            /* number of columns of X (width of the staged tile)                   */
  int32_t n_wide_fixed;       /* leading wide fields with len == 1                                   */
  int32_t max_dim;            /* max dim over deep fields (0 if none)                                */
  int32_t vec;                /* 4, 2 or 1: every deep dim, out_off, ld and base pointer is a        */
                              /* multiple of `vec` floats -- lets rows move as dwordx4/x2            */
  int32_t flags;              /* DCTR_PLAN_* : facts about the device arrays the host cannot see     */
  int32_t* step_sync;         /* nullable: dctr_embed_fwd signals DCTR_SYNC_GATHER in this block when its outputs  */
                              /* have left the chip's caches (see dctr_step_wait)                                  */
  const uint64_t* out_chunks; /* nullable, device: dctr_embed_fwd writes output row b (and its wide logit, which must lie   */
  int32_t chunk_rows;         /* inside the row: wide = out + k, ld_wide = ld_out) to (float*)out_chunks[b / chunk_rows] +  */
  int32_t pad_;               /* (b % chunk_rows) * ld_out instead of out + b * ld_out: the owner's gather of the sharded   */
                              /* step pushes every rank's rows straight into that rank's receive buffer (peer memory)      */
  const dctr_plan_ext_t* ext; /* nullable (host): general update units -- pooled VarLen fields, shared tables (above)     */
} dctr_plan_t;

#define DCTR_PLAN_HAS_GACC 1    /* every field has a gacc slab                       */
#define DCTR_PLAN_HAS_STATE 2   /* every field has an optimizer state slab           */
#define DCTR_PLAN_HAS_MAXPOOL 4 /* some VarLen field uses DCTR_POOL_MAX              */

int dctr_abi_version(void);
const char* dctr_strerror(int code);
size_t dctr_sizeof_field(void); /* binding self-check */
size_t dctr_sizeof_plan(void);

/* ---- fused multi-table lookup (+VarLen pooling, +wide logit, +FM, +DNN-input layout) -------------
 * Replaces, in ONE launch: BaseModel.input_from_feature_columns (basemodel.py:354-380),
 * varlen_embedding_lookup + get_varlen_pooling_list + SequencePoolingLayer (inputs.py:141-155,213-227,
 * sequence.py:49-77), Linear.forward (basemodel.py:63-92), FM.forward (interaction.py:26-34) and
 * combined_dnn_input (inputs.py:126-138).
 *   X     [B, ldx]   the model input matrix (ids as float32)
 *   out   [B, ld_out] row b = [ deep field slices at field.out_off | dense block at plan.dense_off ]
 *   wide  [B]  sum_f w_f[id] (+pooled VarLen) + dense . Linear.weight     (nullable); element b lives at
 *         wide[b * ld_wide] (ld_wide = 1 for a plain vector; a column of `out` for the sharded exchange)
 *   fm    [B]  0.5 * sum_d ((sum_f e)^2 - sum_f e^2) over ALL deep fields (nullable; needs emb_dim)
 *   err   int32 flag; bit0 is set when an id falls outside [0, vocab) -- such a row reads as row 0
 *         (the reference raises IndexError on CPU; here the flag is polled by the host) (nullable)
 * A plan with general units (plan->ext) takes its ids / tags from dctr_embed_ids instead (ids_t must be NULL here); when
 * ext->amax is set the launch writes, for every max-pooled field, the position of the first maximum per element
 * (uint8 [B, ext->ld_amax] at ext->am_deep_off[f] / am_wide_off[f]): max pooling's backward in dctr_embed_update.
 * Optional side outputs feed dctr_embed_update (all nullable):
 *   ids_t [n_units, B] int32: ids_t[u][b] = (int) X[b, units[u].col]  (units: see dctr_embed_update)
 *   parts_t [n_units, B] uint16: clamp(ids_t[u][b]) mod dctr_embed_update_partitions(plan, B)  (needs ids_t)
 *   fm_s  [B, ld_s]    S[b, d] = sum_f e[b, f, d], the per-sample field sum FM's backward needs      */
int dctr_embed_fwd(const dctr_plan_t* plan, const float* X, int64_t ldx, int32_t B, float* out,
                   int64_t ld_out, float* wide, int64_t ld_wide, float* fm, int32_t* err,
                   const int32_t* units, int32_t n_units, int32_t* ids_t, uint16_t* parts_t, float* fm_s,
                   int64_t ld_s, dctr_stream_t stream);

/* ---- backward of the above = embedding_dense_backward + FM backward, as an O(batch) scatter -------
 * Replaces autograd's aten::embedding_dense_backward x(n_deep+n_wide), the pooling backward and
 * FM's backward (called from basemodel.py:261).  Duplicate ids add (atomics).
 *   g_out  [B, ld_g]  gradient w.r.t. `out` (deep slices are read; dense block ignored)  (nullable)
 *   out    [B, ld_out] the forward output (read only when g_fm != NULL: FM backward needs e and sum e)
 *   g_fm   [B]  gradient w.r.t. fm          (nullable)
 *   g_wide [B]  gradient w.r.t. wide        (nullable)
 *   mode   DCTR_BWD_ACCUM: field.gacc[row] += g          (exact dense-gradient semantics, or pass 1
 *                          of a two-pass sparse optimizer)
 *          DCTR_BWD_SGD  : field.table[row] -= lr * g    (torch.optim.SGD(lr), basemodel.py:449-450;
 *                          only legal without max-pooled fields, whose backward re-reads the table) */
#define DCTR_BWD_ACCUM 0
#define DCTR_BWD_SGD 1
int dctr_embed_bwd(const dctr_plan_t* plan, const float* X, int64_t ldx, int32_t B,
                   const float* g_out, int64_t ld_g, const float* out, int64_t ld_out,
                   const float* g_fm, const float* g_wide, int32_t mode, float lr,
                   dctr_stream_t stream);

/* ---- pass 2 of the sparse optimizers: consume + re-zero gacc rows touched by this batch -----------
 * For every id occurrence in X: G = exchange(gacc[row], 0); then per element with G != 0
 *   DCTR_OPT_SGD     p -= lr * G                                   (torch.optim.SGD, basemodel.py:450)
 *   DCTR_OPT_ADAGRAD s += G*G ; p -= lr * G / (sqrt(s) + eps)      (torch.optim.Adagrad, :454)
 * Rows whose gradient is zero do not move under these two optimizers, so the result equals the
 * reference's dense update.                                                      */
#define DCTR_OPT_SGD 0
#define DCTR_OPT_ADAGRAD 1
int dctr_embed_apply(const dctr_plan_t* plan, const float* X, int64_t ldx, int32_t B, int32_t opt,
                     float lr, float eps, dctr_stream_t stream);

/* ---- deterministic fused backward + optimizer (csrc/update.hip) ------------------------------------
 * The O(batch) replacement of embedding_dense_backward + the pooling's backward + FM backward + the optimizer's walk
 * over the tables (basemodel.py:261-262).  Plans of fixed-length fields over distinct tables run one-column units
 * (`units` below); pooled VarLen fields and tables shared through embedding_name run GENERAL units (plan->ext,
 * dctr_plan_ext_t above: then n_units = ext->n_vunits, ids_t / parts_t are [ext->n_vcols, B], parts_t is required, `out`
 * must be the forward's rows when g_fm is given, ext->den_t / ext->amax must point at this step's side buffers).  No atomics:
 * workgroup (unit, partition) owns rows {id : id mod P == partition}, sorts its (id, b) entries in LDS
 * and read-modify-writes each touched row exactly once, summing duplicate ids in (id, b) order, so the
 * result is bit-reproducible (needed for replica equality under data parallelism).
 *   units  [n_units][4] int32 (device): {deep field index | -1, wide field index | -1, X column, 0};
 *          a unit is one id column with the deep and/or wide table it feeds
 *   ids_t  [n_units, B] int32 from dctr_embed_fwd / dctr_embed_ids;  parts_t [n_units, B] uint16 from the same
 *          call (nullable: every workgroup then divides every id of its unit itself)
 *   g_out / out / fm_s / g_fm / g_wide as in dctr_embed_bwd (fm_s = side output of dctr_embed_fwd);
 *          g_wide[b] lives at g_wide[b * ld_gw]
 *   opt    DCTR_UPD_SGD      table[row] -= lr * G                          (torch.optim.SGD)
 *          DCTR_UPD_ADAGRAD  state[row] += G*G ; table[row] -= lr*G/(sqrt(state[row])+eps)
 *          DCTR_UPD_ACCUM    gacc[row]  += G          (exact dense-gradient semantics: param.grad)
 *   max_vocab  largest vocab over the plan's fields (sizes the 32-bit sort keys)
 *   g_wdense [plan.n_wdense] (nullable): when given, n_wdense extra workgroups also write the gradient of the dense
 *          half of Linear, g_wdense[j] = sum_b g_wide[b] * X[b, wdense_cols[j]] (basemodel.py:86-90), in a
 *          fixed order; X / ld_x are only read for this.  wdense_step (nullable): they also step Linear.weight.
 * dctr_embed_update_supported returns 1 when the plan / batch fit (B <= 2^20, keys fit 32 bits).    */
#define DCTR_UPD_SGD 0
#define DCTR_UPD_ADAGRAD 1
#define DCTR_UPD_ACCUM 2
#define DCTR_UPD_LAZY 3   /* dctr_embed_update_lazy only: the lazily regularised / Adam step at the row (see below) */
/* One optimizer step on dense parameters, applied by the kernel that finishes their gradient (instead of a separate
 * dctr_dense_opt launch and the cross-queue join in front of it): the gradient tensor lives at grad_base + k inside
 * a flat gradient slab, its parameter at param_base + k, its Adagrad state at state_base + k.
 *   kind  DCTR_UPD_SGD: p -= lr * g        DCTR_UPD_ADAGRAD: s += g*g ; p -= lr * g / (sqrt(s) + eps)
 * (torch.optim.SGD / Adagrad, basemodel.py:447-461).  Host struct.                                                */
typedef struct dctr_dense_step {
  int32_t kind;
  float lr, eps;
  int32_t pad_;
  const float* grad_base;
  float* param_base;
  float* state_base; /* NULL for SGD */
} dctr_dense_step_t;
size_t dctr_sizeof_dense_step(void);

int dctr_embed_update_supported(const dctr_plan_t* plan, int64_t max_vocab, int32_t B);
/* P = partitions per unit that dctr_embed_update uses for this plan at batch B (0 on bad arguments).  The optional
 * side output parts_t [n_units, B] uint16 of dctr_embed_fwd / dctr_embed_ids holds clamp(id) mod P (clamp: an id
 * outside [0, vocab) counts as 0): with it a workgroup's scan over its unit's B entries is a 16-bit compare per
 * entry instead of an integer division per entry.                                                              */
int32_t dctr_embed_update_partitions(const dctr_plan_t* plan, int32_t B);
#ifdef DCTR_DIAG
/* diagnostics, only in the DCTR_DIAG build (make -C csrc diag -> libdctr_hip_diag.so; tools/upd_trace.py): buf != NULL
 * makes every workgroup of dctr_embed_update record 8 u64 (wall_clock64 at start / scan / sort / loads issued / loads
 * landed / tile 0 done / end, then its entry count); force_p > 0 overrides the number of partitions per unit.
 * dctr_dbg_update_trace(NULL, -1) restores normal operation.  The shipped library has no mutable global state.   */
void dctr_dbg_update_trace(unsigned long long* buf, int32_t force_p);
#endif
/* plan is only needed (non-NULL) when parts_t is requested.  A plan with general units (plan->ext): n_units =
 * ext->n_vunits, ids_t / parts_t are [ext->n_vcols, B] (one row per X column feeding a unit; a position that sum / mean
 * pooling masks out gets the tag 0xFFFF), and ext->den_t [ext->n_den, B] receives mean pooling's divisors count + 1e-8. */
int dctr_embed_ids(const dctr_plan_t* plan, const int32_t* units, int32_t n_units, const float* X, int64_t ldx,
                   int32_t B, int32_t* ids_t, uint16_t* parts_t, dctr_stream_t stream);
int dctr_embed_update(const dctr_plan_t* plan, const int32_t* units, int32_t n_units, int64_t max_vocab,
                      const int32_t* ids_t, const uint16_t* parts_t, int32_t B, const float* g_out, int64_t ld_g,
                      const float* out, int64_t ld_out, const float* fm_s, int64_t ld_s, const float* g_fm,
                      const float* g_wide, int64_t ld_gw, int32_t opt, float lr, float eps, const float* X,
                      int64_t ld_x, float* g_wdense, const dctr_dense_step_t* wdense_step, int32_t* workspace,
                      int64_t workspace_ints, int32_t presorted, dctr_stream_t stream);
/* The part of dctr_embed_update that needs nothing but the ids, as its own launch: every (unit, partition)'s entries
 * found, sorted by (id, sample) and parked in `workspace` (dctr_embed_update_workspace_ints ints; zero before the
 * first use).  Enqueue it any time after the forward -- on another stream, in the shadow of the tower -- and pass the
 * same workspace with presorted = 1 to dctr_embed_update: its workgroups then start with one coalesced read of their
 * keys (no scan, no sort).  Same results, bit for bit.  dctr_embed_update leaves the workspace ready for the next
 * dctr_embed_segments.                                                                                         */
int dctr_embed_segments(const dctr_plan_t* plan, const int32_t* units, int32_t n_units, int64_t max_vocab,
                        const int32_t* ids_t, const uint16_t* parts_t, int32_t B, int32_t* workspace,
                        int64_t workspace_ints, dctr_stream_t stream);
/* workspace (nullable): dctr_embed_update_workspace_ints(plan, n_units, B) int32, ZERO before the first use (the
 * kernels leave it ready for the next launch).  With it (and presorted = 0) a pre-pass buckets the (unit, sample)
 * entries by partition, so no workgroup scans a unit's B ids: worth it for large (global) batches; without it every
 * workgroup scans.  presorted = 1: see dctr_embed_segments.                                                     */
int64_t dctr_embed_update_workspace_ints(const dctr_plan_t* plan, int32_t n_units, int32_t B);

/* ---- FM on an explicit [B, F, D] tensor (interaction.py:26-34) ------------------------------------
 * E is addressed as E[b*ld_b + f*D + d].  y[b] = 0.5 * sum_d((sum_f e)^2 - sum_f e^2).
 * backward: gE[b,f,d] (+)= gy[b] * (S[b,d] - E[b,f,d]);  accumulate != 0 adds into gE.             */
int dctr_fm_fwd(const float* E, int64_t ld_b, int32_t B, int32_t F, int32_t D, float* y,
                dctr_stream_t stream);
int dctr_fm_bwd(const float* E, int64_t ld_b, int32_t B, int32_t F, int32_t D, const float* gy,
                float* gE, int64_t ld_gb, int32_t accumulate, dctr_stream_t stream);

/* ---- BiInteractionPooling (interaction.py:54-61) + NFM's combined_dnn_input (nfm.py:66-71) ---------------------
 *   bi[b, d] = 0.5 * ((sum_f e[b,f,d])^2 - sum_f e[b,f,d]^2)           out row b = [ bi (D) | dense (n_dense) ]
 * G is dctr_embed_fwd's `out` ([B, ld_g], fields first, the dense block at dense_off); the backward writes a gradient
 * with G's layout (what dctr_embed_update consumes): gG[b, f*D+d] = g[b,d] * (S[b,d] - e[b,f,d]), gG[b, dense_off+j]
 * = g[b, D+j].  Columns of gG outside the field / dense blocks are not written.                                  */
int dctr_bi_pooling_fwd(const float* G, int64_t ld_g, int32_t B, int32_t F, int32_t D, int32_t dense_off,
                        int32_t n_dense, float* out, int64_t ld_o, dctr_stream_t stream);
int dctr_bi_pooling_bwd(const float* G, int64_t ld_g, int32_t B, int32_t F, int32_t D, int32_t dense_off,
                        int32_t n_dense, const float* gout, int64_t ld_go, float* gG, int64_t ld_gg,
                        dctr_stream_t stream);

/* ---- AFMLayer (interaction.py:251-325): attentional pooling of the pairwise products (csrc/afm.hip) -----------
 *   bi_k = e_i (.) e_j (pairs i < j, itertools.combinations order);  t_k = relu(bi_k W + bias);  s_k = t_k . h;
 *   a = softmax_k(s);  out = sum_k a_k bi_k;  y[b] = out . p
 *   E [B, F, D] rows at E + b*ld_e;  W [D, A] (attention_W), bias [A] (attention_b), h [A] (projection_h), p [D]
 *   (projection_p).  One wave per sample, everything between E and y stays in LDS.  The backward recomputes the
 *   forward, writes gE [B, F*D] rows at gE + b*ld_ge and the four parameter gradients (fixed-order sums, no
 *   atomics); workspace: dctr_afm_bwd_workspace_floats(B, D, A) floats.  Needs D <= 64, A <= 32, F <= 64.      */
size_t dctr_afm_bwd_workspace_floats(int32_t B, int32_t D, int32_t A);
int dctr_afm_fwd(const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, int32_t A, const float* W,
                 const float* bias, const float* h, const float* p, float* y, dctr_stream_t stream);
int dctr_afm_bwd(const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, int32_t A, const float* W,
                 const float* bias, const float* h, const float* p, const float* gy, float* gE, int64_t ld_ge,
                 float* gW, float* gbias, float* gh, float* gp, float* workspace, dctr_stream_t stream);

/* ---- InteractingLayer of AutoInt (interaction.py:328-394): multi-head self-attention over the fields (csrc/interact.hip)
 *   Q = E Wq, K = E Wk, V = E Wv;  head n = columns [n*A, (n+1)*A), A = D / H;  P_n = softmax_rows(Q_n K_n^T (/ sqrt(A)
 *   when scaling));  out = relu([P_0 V_0 | ... | P_{H-1} V_{H-1}] + E Wr)          (Wr = NULL: no residual)
 *   E [B, F, D] rows at E + b*ld_e;  W* [D, D] row-major (x @ W);  out [B, F*D] rows at out + b*ld_o.
 * One wave per sample, everything between E and out stays in LDS; the backward recomputes the forward, writes gE and
 * the four weight gradients (fixed-order sums, no atomics); workspace: dctr_interacting_bwd_workspace_floats(B, D).
 * dctr_interacting_supported(F, D, H) != 0 iff the shape fits (D <= 32, F <= 64, LDS).                               */
int dctr_interacting_supported(int32_t F, int32_t D, int32_t H);
size_t dctr_interacting_bwd_workspace_floats(int32_t B, int32_t D);
int dctr_interacting_fwd(const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, int32_t H, int32_t scaling,
                         const float* Wq, const float* Wk, const float* Wv, const float* Wr, float* out, int64_t ld_o,
                         dctr_stream_t stream);
int dctr_interacting_bwd(const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, int32_t H, int32_t scaling,
                         const float* Wq, const float* Wk, const float* Wv, const float* Wr, const float* gout,
                         int64_t ld_g, float* gE, int64_t ld_ge, float* gWq, float* gWk, float* gWv, float* gWr,
                         float* workspace, dctr_stream_t stream);

/* ---- lazy regularised / Adam embedding update (csrc/lazy.hip) -------------------------------------------------
 * Replaces, without the O(vocabulary) memory traffic, what the reference does whenever every row of a table moves at
 * every step: the dense L2 gradient 2*lambda*w of get_regularization_loss (basemodel.py:412-428; l2_reg_embedding /
 * l2_reg_linear default to 1e-5) and torch.optim.Adam's moment-driven updates of untouched rows
 * (basemodel.py:447-461).  A row's trajectory between two batches that touch it depends on the row alone, so it is
 * replayed -- the same recurrence, step by step -- when the row is next needed.  One "unit" = one id column of X with
 * the deep and / or wide table it feeds (as in dctr_embed_update); `stamp[row]` = optimizer steps already applied
 * to the row, `*step` (device) = steps completed so far.  Per train step the caller enqueues
 *   dctr_embed_ids -> dctr_lazy_catchup -> dctr_embed_fwd -> ... -> dctr_embed_update(DCTR_UPD_ACCUM)
 *   -> dctr_lazy_apply -> dctr_lazy_step_inc,
 * and dctr_lazy_flush before anything else reads the tables (predict / evaluate / state_dict).
 * dctr_lazy_sweep (optional, once per step) goes BEHIND the catch-up -- never beside it: sweep and catch-up read and write
 * the stamps non-atomically and skip a row only when its stamp is already current, so a sweep that overlaps a catch-up
 * replays rows twice -- and must have finished before dctr_lazy_step_inc (it reads the step counter).  Between the two it
 * may run on a stream of its own beside the gather, the tower, the update and dctr_lazy_apply: they touch the batch's
 * rows, which the catch-up has stamped current and the sweep therefore skips.
 *   *_s1  Adagrad `sum` | Adam `exp_avg`     *_s2  Adam `exp_avg_sq`     *_g  gradient slab, zero at rest
 *   vec   1 or 4: every deep dim and base pointer is a multiple of `vec` floats;  max_dim <= 64*vec                */
#define DCTR_LAZY_SGD 0
#define DCTR_LAZY_ADAGRAD 1
#define DCTR_LAZY_ADAM 2
#define DCTR_LAZY_RMSPROP 3 /* torch.optim.RMSprop (momentum 0, not centered): s1 = square_avg, beta2 = alpha,  */
                            /* beta1 = 1 - alpha.  A row no sample refers to still has its square_avg decayed at */
                            /* every step: replayed like Adam's moments                                          */
typedef struct dctr_lazy_unit {
  float* deep;    /* [vocab, dim] (rows ld_deep floats apart) or NULL */
  float* deep_s1; /* rows ld_deep_s1 floats apart; deep_s2 / deep_g are contiguous */
  float* deep_s2;
  float* deep_g;
  float* wide;    /* [vocab] or NULL */
  float* wide_s1;
  float* wide_s2;
  float* wide_g;
  int32_t* stamp; /* [vocab] */
  int64_t vocab;
  int32_t dim;
  int32_t col;
  float l2_deep;  /* lambda of the L2 term lambda * sum(w^2) on the deep table */
  float l2_wide;
  int32_t ld_deep, ld_deep_s1; /* row strides in floats, 0 = dim */
  int32_t ld_wide, ld_wide_s1; /* row strides in floats, 0 = 1   */
} dctr_lazy_unit_t;
typedef struct dctr_lazy_opt {
  int32_t kind; /* DCTR_LAZY_* */
  float lr, eps, beta1, beta2;
  /* Adam only, optional (NULL: the kernels compute the scalars themselves, in double, per step and lane): DEVICE tables of
   * the step-dependent scalars torch.optim.Adam computes on the host (adam.py: bias_correction = 1 - beta ** step),
   *   adam_ss[T - 1] = (float)(lr / (1 - beta1^T))     adam_bc[T - 1] = (float)sqrt(1 - beta2^T)      T = 1 .. n
   * each long enough that its last entry is the limit (1 - beta^T == 1 in double: 349 / 36 708 steps at the default
   * betas): steps past the end read the last entry.  A row that slept k steps replays k optimizer steps in its catch-up:
   * the tables take two double multiplies, a division and a square root per replayed step and lane out of that loop.   */
  int32_t n_ss, n_bc;
  int32_t any_l2; /* some unit carries an L2 term (SGD / Adagrad rows then move between two touches: the catch-up has steps
                     to replay and orders its entries by gap) */
  const float* adam_ss;
  const float* adam_bc;
  const float* adam_rbc; /* optional (ABI 24), n_bc entries: (float)(1 / sqrt(1 - beta2^T)) -- the replay loop divides by
                            adam_bc[T - 1] through this reciprocal; NULL: it takes the hardware reciprocal of adam_bc itself,
                            one quarter-rate instruction per replayed step and wavefront */
} dctr_lazy_opt_t;
size_t dctr_sizeof_lazy_unit(void);
/* Round 6 -- the step that carries the batch's DATA gradient, inside the sorted update: dctr_embed_update's gradient sums
 * (same arguments, same order of additions) meet g = G + 2 lambda w and ONE optimizer step of `lazy_opt` on (w, s1, s2) at the
 * row, which is stamped t + 1 (*step = t is read, not advanced: dctr_lazy_step_inc follows as before).  Replaces
 * dctr_embed_update(DCTR_UPD_ACCUM) + dctr_lazy_apply -- no gradient slab, no second pass over the batch's rows.  Needs what
 * csrc/lazy.hip needs (simple units, the batch's rows caught up to t by dctr_lazy_catchup) and pre-sorted entries
 * (dctr_embed_segments on the same ids: `workspace`); DCTR_ENOSUP otherwise -- the caller keeps the two-pass route.       */
int dctr_embed_update_lazy(const dctr_plan_t* plan, const int32_t* units, int32_t n_units, int64_t max_vocab,
                           const int32_t* ids_t, const uint16_t* parts_t, int32_t B, const float* g_out, int64_t ld_g,
                           const float* out, int64_t ld_out, const float* fm_s, int64_t ld_s, const float* g_fm,
                           const float* g_wide, int64_t ld_gw, const float* X, int64_t ld_x, float* g_wdense,
                           int32_t* workspace, int64_t workspace_ints, const struct dctr_lazy_unit* lazy_units,
                           const int32_t* step, const struct dctr_lazy_opt* lazy_opt, dctr_stream_t stream);
size_t dctr_sizeof_lazy_opt(void);
/* order_ws (nullable): n_units * B ints of scratch -- with it the catch-up first deals every unit's entries by the number of
 * steps their rows slept (one launch, a counting sort per unit), so that the rows one wavefront replays side by side need
 * about the same number of steps: geometric gaps otherwise leave a wave waiting for its longest row (3.4 x the mean of 16). */
int dctr_lazy_catchup(const dctr_lazy_unit_t* units, int32_t n_units, const int32_t* ids_t, int32_t B,
                      const int32_t* step, const dctr_lazy_opt_t* opt, int32_t vec, int32_t max_dim,
                      int32_t* order_ws, dctr_stream_t stream);
int dctr_lazy_apply(const dctr_lazy_unit_t* units, int32_t n_units, const int32_t* ids_t, int32_t B,
                    const int32_t* step, const dctr_lazy_opt_t* opt, int32_t vec, int32_t max_dim,
                    dctr_stream_t stream);
int dctr_lazy_flush(const dctr_lazy_unit_t* units, int32_t n_units, int64_t max_vocab, const int32_t* step,
                    const dctr_lazy_opt_t* opt, int32_t vec, int32_t max_dim, dctr_stream_t stream);
/* dctr_lazy_sweep: bring the (*step mod K)-th of K windows of every table's rows ([w * ceil(vocab / K), ...)) to the current
 * step -- the flush of one K-th of the rows, called once per train step in front of the catch-up.  No row then sleeps
 * longer than K steps: a row's replay is a sequential loop over its missed steps, so without the sweep a rarely drawn id
 * stalls the step it finally appears in for its whole history (one lane group, gap x ~0.5 us), and a flush after N steps
 * pays N steps for every row nobody drew.  The arithmetic is the same and each (row, step) is still applied exactly once,
 * in order: results are unchanged.  The window follows the device-side step counter: replayable from a hipGraph.           */
int dctr_lazy_sweep(const dctr_lazy_unit_t* units, int32_t n_units, int64_t max_vocab, int32_t K, const int32_t* step,
                    const dctr_lazy_opt_t* opt, int32_t vec, int32_t max_dim, dctr_stream_t stream);
int dctr_lazy_step_inc(int32_t* step, dctr_stream_t stream);
/* The same optimizer step (number *step + 1) over a flat dense slab with an optional per-element lambda
 * (g = grad + 2*lambda*p): what torch.optim.SGD / Adagrad / Adam do to the dense parameters when the model carries
 * L2 terms (l2_reg_linear on Linear.weight, l2_reg_dnn) -- one launch, no dense autograd node for the L2 term.  */
int dctr_dense_opt_reg(float* p, const float* g, float* s1, float* s2, const float* lam, int64_t n,
                       const dctr_lazy_opt_t* opt, const int32_t* step, dctr_stream_t stream);

/* ---- CIN layer (interaction.py:207-248) on fp32 MFMA (csrc/cin.hip) -------------------------------
 * One Compressed-Interaction layer without ever materialising Z = H (x) X0:
 *     Y[b, o, d] = sum_{h, m} W[o, h*M + m] * H[b, h, d] * X0[b, m, d] + bias[o];   A = relu(Y) if relu
 *   H    [B, h, D]  rows at H  + b*ld_h   (previous layer's "next_hidden", or X0 for layer 0)
 *   X0   [B, M, D]  rows at X0 + b*ld_x0  (the field embeddings; a view of dctr_embed_fwd's `out`)
 *   W    [O, h*M]   conv1ds.<k>.weight with its trailing 1 squeezed;  bias [O] (nullable)
 *   A    [B, O, D]  rows at A + b*ld_a
 *   workspace  dctr_cin_workspace_floats(h, M, O) floats (the kernel's re-laid-out copy of W)
 * Needs M <= 32.  The split_half / sum over d / concat of the reference stay with the caller.
 * Backward, given gA = d loss / d A (and the saved A when relu):
 *   gH   [B, h, D]  (written)      gX0 [B, M, D] (accumulated: += when accumulate_x0 != 0)
 *   gW   [O, h*M]   (written)      gbias [O]     (written, nullable)
 * gA and A share the leading dimension ld_a.  The backward needs dctr_cin_bwd_workspace_floats(B, h, M, D, O)
 * floats of scratch: the weight gradient is a [O, h*M] x (B*D) GEMM whose batch reduction is split over
 * workgroups; their partial tiles go to the workspace and are summed in a fixed order (no atomics).     */
size_t dctr_cin_workspace_floats(int32_t h, int32_t M, int32_t O);
size_t dctr_cin_bwd_workspace_floats(int32_t B, int32_t h, int32_t M, int32_t D, int32_t O);
int dctr_cin_layer_fwd(const float* H, int64_t ld_h, const float* X0, int64_t ld_x0, const float* W,
                       const float* bias, int32_t B, int32_t h, int32_t M, int32_t D, int32_t O, int32_t relu,
                       float* A, int64_t ld_a, float* workspace, dctr_stream_t stream);
int dctr_cin_layer_bwd(const float* gA, const float* A, int64_t ld_a, int32_t relu, const float* H, int64_t ld_h,
                       const float* X0, int64_t ld_x0, const float* W, int32_t B, int32_t h, int32_t M, int32_t D,
                       int32_t O, float* gH, int64_t ld_gh, float* gX0, int64_t ld_gx, int32_t accumulate_x0,
                       float* gW, float* gbias, float* workspace, dctr_stream_t stream);

/* The layer's "direct connect" rows as xDeepFM consumes them (interaction.py:226-246: split / cat / sum(-1)):
 *   dctr_cin_pool_fwd: pooled[b * ld_pooled + o] = sum_d A[b, pool_from + o, d]   (A [B, O, D] contiguous; `pooled` may be
 *                      the layer's block of the CIN's [B, featuremap_num] output: no torch.cat behind the layers)
 *   dctr_cin_pool_bwd: gA[b, o, :] = (o < n_hidden ? g_hidden[b, o, :] : 0) + (o >= pool_from ? gp(b, o - pool_from) : 0)
 *                      g_hidden [B, n_hidden, D] contiguous (NULL = 0); gp(b, j) = g_pooled[b * ld_gp + j] (NULL = 0), or,
 *                      with w_head: g_pooled[b * ld_gp] * w_head[j] -- the backward of the bias-free 1-unit Linear xDeepFM
 *                      puts on the CIN output (xdeepfm.py:72, :97) folded in: g_pooled is then the logit's gradient.
 *                      split_half: pool_from == n_hidden (disjoint row sets); otherwise pool_from = 0 and every row below
 *                      n_hidden receives both terms (interaction.py:240-242).
 *                      A_relu (nullable, [B, O, D]): the layer's saved relu output -- gA is zeroed where it is not > 0, i.e.
 *                      the relu's backward is applied HERE and dctr_cin_layer_bwd is then called with relu = 0 (its two
 *                      kernels stage 16 gradient rows per memory round trip instead of 8 + 8 mask rows)               */
int dctr_cin_pool_fwd(const float* A, int32_t B, int32_t O, int32_t D, int32_t pool_from, float* pooled,
                      int64_t ld_pooled, dctr_stream_t stream);
int dctr_cin_pool_bwd(const float* g_hidden, const float* g_pooled, int64_t ld_gp, const float* w_head,
                      const float* A_relu, int32_t B, int32_t O, int32_t D, int32_t n_hidden, int32_t pool_from,
                      float* gA, dctr_stream_t stream);

/* out[b] = sum_j x[b * ld_x + j] * w[j]   (csrc/head.hip): nn.Linear(N, 1, bias=False) over narrow rows -- xDeepFM's
 * cin_linear (xdeepfm.py:72, :97).  One wave per row, fixed summation order.                                          */
int dctr_rows_dot(const float* x, int64_t ld_x, const float* w, int32_t B, int32_t N, float* out, dctr_stream_t stream);
/* out[j] = sum_b w[b] * x[b * ld_x + j]   (ABI 24): the weight gradient of that projection, g_w = g^T X -- torch.mm of a
 * [1, B] row with the [B, N] feature maps before (a 16 us library GEMM at B = 4096, N = 192).  Rows in groups of 32, the
 * groups' sums added in group order: deterministic.  workspace: dctr_relu_bwd_bias_workspace_floats(B, N) floats.   */
int dctr_rows_tdot(const float* x, int64_t ld_x, const float* w, int32_t B, int32_t N, float* out, float* workspace,
                   dctr_stream_t stream);

/* ---- SENET / Bilinear / InnerProduct (csrc/pairwise.hip) ----------------------------------------------
 * SENETLayer (interaction.py:93-101): z = mean_d E; a1 = relu(z W1^T); a = relu(a1 W2^T); V = E * a[:, :, None]
 *   E [B, F, D] rows at E + b*ld_e;  W1 [R, F], W2 [F, R] (excitation.0 / .2 weights);  V [B, F*D] contiguous;
 *   a [B, F] and a1 [B, R] are saved for the backward.  Backward writes gE [B, F*D], gW1, gW2
 *   (workspace: dctr_senet_bwd_workspace_floats(B, F, R) floats of per-workgroup partials, summed in fixed order). */
int dctr_senet_fwd(const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, const float* W1, const float* W2,
                   int32_t R, float* V, float* a, float* a1, dctr_stream_t stream);
size_t dctr_senet_bwd_workspace_floats(int32_t B, int32_t F, int32_t R);
int dctr_senet_bwd(const float* gV, const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, const float* W1,
                   const float* W2, int32_t R, const float* a, const float* a1, float* gE, float* gW1, float* gW2,
                   float* workspace, dctr_stream_t stream);

/* BilinearInteraction (interaction.py:140-156): p_k = (x_i W_w^T) (.) x_j for the pairs k = (i, j), i < j.
 *   Wf    [n_w, D, D]  the nn.Linear weights stacked: n_w = 1 ("all"), F ("each", w = i), P ("interaction", w = k)
 *   sched [n_sched][4] int32 (device): {i, j, w, k}.  The FORWARD takes the entries in output order (k ascending, no
 *         idle entries, n_sched = P): the waves of a workgroup then write neighbouring pieces of a sample's row
 *         together.  The BACKWARD takes two tables: `sched` in round-robin-tournament order, `slots` entries per round,
 *         i = -1 for an idle slot (its data kernel lets four waves accumulate into LDS rows without conflicts: a round
 *         is a perfect matching of the fields), and `sched_k` in output order like the forward's (its weight kernel
 *         gives every wave 8 consecutive pairs, i.e. whole lines of the incoming gradient; NULL: `sched` is used).
 *         pair_w [P] int32: weight index of pair k
 *   V     optional second input (FiBiNET's SENET output, fibinet.py:82-83).  With V: out row =
 *         [ V pairs (P*D) | E pairs (P*D) ], else [ E pairs ].  `dense` (nullable): n_dense floats per sample copied
 *         to out[:, dense_off ...) so that `out` IS the reference's DNN input (fibinet.py:86-87).  D <= 16.
 *   backward: gout has out's layout; writes gE (and gV) [B, F*D] and gW [n_w, D, D];
 *         workspace = dctr_bilinear_bwd_workspace_floats(B, P, D) floats.                                     */
int dctr_bilinear_fwd(const float* E, int64_t ld_e, const float* V, int64_t ld_v, const float* Wf,
                      const int32_t* sched, int32_t n_sched, int32_t P, int32_t F, int32_t D, int32_t B, float* out,
                      int64_t ld_o, const float* dense, int64_t ld_d, int32_t n_dense, int32_t dense_off,
                      dctr_stream_t stream);
size_t dctr_bilinear_bwd_workspace_floats(int32_t B, int32_t P, int32_t D);
int dctr_bilinear_bwd(const float* E, int64_t ld_e, const float* V, int64_t ld_v, const float* Wf,
                      const int32_t* sched, int32_t n_sched, int32_t slots, const int32_t* pair_w, int32_t n_w,
                      int32_t P, int32_t F, int32_t D, int32_t B, const float* gout, int64_t ld_g, float* gE,
                      float* gV, float* gW, float* workspace, const int32_t* sched_k, int32_t n_sched_k,
                      dctr_stream_t stream);

/* FiBiNET's bilinear pairs TOGETHER WITH the first tower layer behind them, backward direction (fibinet.py:82-99:
 * dnn_input = [ Bilinear(senet) | Bilinear(raw) | dense ], h1 = act(W0 dnn_input + b0); core.py:123-133).  Replaces
 * `gout = gh W0` (a [B, 2 P D] slab) followed by dctr_bilinear_bwd: the gradient of a (16 samples x one pair) piece is
 * produced on the matrix cores where it is consumed, the slab is neither written nor read.
 *   gh     [B, H] at gh + b*ld_gh: gradient at the first layer's pre-activation (after the activation's backward)
 *   W0     [H, >= 2 P D] at W0 + h*ld_w0: the nn.Linear weight; columns [0, P D) belong to V's pairs, [P D, 2 P D) to E's
 *   sched4 [n_groups][8][4] int32 (device): groups of eight field-disjoint pairs {i, j, w, k} (one per wave of a
 *          512-thread workgroup), i = -1 for an idle entry
 *   writes gE, gV [B, F*D] and gW [n_w, D, D]; workspace = dctr_bilinear_wide_bwd_workspace_floats(B, P) floats.
 *   DCTR_ENOSUP unless D == 16, n_w == P (one weight per pair: "interaction"), H <= 128, H % 4 == 0, 16-byte aligned rows. */
size_t dctr_bilinear_wide_bwd_workspace_floats(int32_t B, int32_t P);
/* The same node, forward direction: h [B, H] = act(W0 x + b0) with the pairs of x made where they are consumed (no
 * [B, 2 P D + n_dense] x [.., H] library GEMM).  x (row stride ld_x, 16-byte aligned rows, ALLOCATED for whole tiles of
 * 32 rows: ceil(B / 32) * 32 rows, the rows past B receive zeros) is still written -- the
 * backward's weight-gradient GEMM reads it -- with bits equal to dctr_bilinear_fwd's.  sched_k: the pairs in output
 * order as dctr_bilinear_fwd takes them; relu != 0: act = relu, else identity; b0 nullable.  The eight per-wave partial
 * sums of a sample's row meet in wave order, the two inputs' shares in input order: bit-reproducible.
 * workspace = dctr_bilinear_wide_fwd_workspace_floats(B, P) floats.  DCTR_ENOSUP unless D == 16, H <= 128, n_dense <= 32.  (fibinet.py:82-99, interaction.py:140-156, core.py:123-133)           */
size_t dctr_bilinear_wide_fwd_workspace_floats(int32_t B, int32_t P);
int dctr_bilinear_wide_fwd(const float* E, int64_t ld_e, const float* V, int64_t ld_v, const float* Wf,
                           const int32_t* sched_k, int32_t P, int32_t F, int32_t D, int32_t B, const float* dense,
                           int64_t ld_d, int32_t n_dense, const float* W0, int64_t ld_w0, int32_t H, const float* b0,
                           int32_t relu, float* x, int64_t ld_x, float* h, int64_t ld_h, float* workspace,
                           float* wpk_bwd, dctr_stream_t stream);
/* wpk_bwd (nullable, dctr_bilinear_wide_pack_floats(P) floats, 16-byte aligned): W0 in the BACKWARD's operand layout,
 * written by the forward's packing launch; handed to dctr_bilinear_wide_bwd as `wpk` (NULL there: it packs W0 itself)
 * it saves that launch.  Valid as long as W0 is unchanged.                                                           */
size_t dctr_bilinear_wide_pack_floats(int32_t P);
int dctr_bilinear_wide_bwd(const float* E, int64_t ld_e, const float* V, int64_t ld_v, const float* Wf,
                           const int32_t* sched4, int32_t n_groups, const int32_t* pair_w, int32_t n_w, int32_t P,
                           int32_t F, int32_t D, int32_t B, const float* gh, int64_t ld_gh, const float* W0,
                           int64_t ld_w0, int32_t H, float* gE, float* gV, float* gW, float* workspace,
                           const float* wpk, dctr_stream_t stream);

/* InnerProductLayer (interaction.py:557-577): out[b, k] = sum_d e_i e_j (reduce != 0) or out[b, k*D + d] = e_i e_j;
 * pair order i < j, i outer.  Backward: gE[b, f, :] = sum_{g != f} gp[b, pair(f, g)] e_g.                    */
int dctr_inner_product_fwd(const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, int32_t reduce, float* out,
                           int64_t ld_o, dctr_stream_t stream);
int dctr_inner_product_bwd(const float* E, int64_t ld_e, int32_t B, int32_t F, int32_t D, int32_t reduce,
                           const float* gp, int64_t ld_g, float* gE, int64_t ld_ge, dctr_stream_t stream);

/* ---- CrossNet, vector parameterisation (interaction.py:438-447; csrc/cross.hip) --------------------------
 *     x_{l+1} = x_0 * (x_l . w_l) + b_l + x_l ,  l = 0..L-1
 *   X [B, W] rows at X + b*ld_x;  kernels [L, W] (crossnet.kernels [L, W, 1]);  bias [L, W];  Y [B, W].  W <= 2048.
 * Backward writes gX [B, W], g_kernels [L, W], g_bias [L, W];
 * workspace = dctr_crossnet_vec_bwd_workspace_floats(B, W, L) floats (per-workgroup partials, fixed-order sum).
 * (The matrix parameterisation: dctr_crossnet_mat_fwd / _bwd below.)                                             */
int dctr_crossnet_vec_fwd(const float* X, int64_t ld_x, int32_t B, int32_t W, int32_t L, const float* kernels,
                          const float* bias, float* Y, int64_t ld_y, dctr_stream_t stream);
size_t dctr_crossnet_vec_bwd_workspace_floats(int32_t B, int32_t W, int32_t L);
int dctr_crossnet_vec_bwd(const float* X, int64_t ld_x, int32_t B, int32_t W, int32_t L, const float* kernels,
                          const float* bias, const float* gY, int64_t ld_g, float* gX, int64_t ld_gx,
                          float* g_kernels, float* g_bias, float* workspace, dctr_stream_t stream);

/* ---- CrossNet, matrix parameterisation (interaction.py:448-451; csrc/mlp.hip, on the tower's MFMA machinery) ------
 *     x_{l+1} = x_0 (.) (x_l W_l^T + b_l) + x_l ,  l = 0..L-1          W_l = crossnet.kernels[l]  [W, W]
 * described as a dctr_mlp_t (declared below) whose layers are all W x W: layer[l].W (rows ld_w floats apart, ld_w % 4
 * == 0), .bias [W], .h [B, ld_h] receives x_{l+1} (the last one IS the result), .dh [B, ld_h] is scratch shared by
 * the two calls (the forward parks u_l = x_l W_l^T + b_l there, the backward turns it into d loss / d u_l);  w_out must
 * be NULL.  A workgroup carries 16 samples through all layers (x_0 and x_l stay in LDS); fp32 MFMA 16x16x4.
 * The backward takes gY = d loss / d x_L [B, ld_g] and writes gx [B, ld_gx] = d loss / d x_0, layer[l].gW [W, ld_w]
 * and layer[l].gbias [W] (split-batch partials summed in a fixed order: no atomics);
 * workspace = dctr_crossnet_mat_bwd_workspace_floats(m, B) floats.  dctr_crossnet_mat_supported: W <= 512 (LDS).   */
struct dctr_mlp;
int dctr_crossnet_mat_supported(int32_t W, int32_t n_layers);
int dctr_crossnet_mat_fwd(const struct dctr_mlp* m, const float* x, int64_t ld_x, int32_t B, dctr_stream_t stream);
size_t dctr_crossnet_mat_bwd_workspace_floats(const struct dctr_mlp* m, int32_t B);
int dctr_crossnet_mat_bwd(const struct dctr_mlp* m, const float* x, int64_t ld_x, int32_t B, const float* gY,
                          int64_t ld_g, float* gx, int64_t ld_gx, float* workspace, dctr_stream_t stream);

/* ---- CrossNetMix of DCN-Mix (interaction.py:499-534; csrc/mlp.hip) -------------------------------------------------
 * Per cross layer, with E experts of rank R over W inputs (G = the gating weights [E, W], shared by all layers):
 *     s = softmax(x_l G^T)                      v1_e = tanh(x_l V_e)                  v2_e = tanh(v1_e C_e^T)
 *     x_{l+1} = x_0 (.) (sum_e s_e v2_e U_e^T + b) + x_l
 * described as a dctr_mlp_t of THREE dense layers per cross layer (the caller packs the weights):
 *   layer 3l   : W1 [E*R + E, W]   rows e*R + r = V_e[:, r], rows E*R + e = G[e];   h = [v1 | s],  dh scratch
 *   layer 3l+1 : W2 [E*R, E*R]     block-diagonal, block e = C_e;   h = s (.) v2 (what feeds the next product),
 *                                  dh: the forward parks the unscaled v2 there (the backward needs it)
 *   layer 3l+2 : W3 [W, E*R]       W3[w, e*R + r] = U_e[w, r];  bias = b;   h = x_{l+1} (the last one IS the result),
 *                                  dh: the forward parks u = sum_e s_e v2_e U_e^T + b
 * ld_w % 4 == 0 with zero padding, w_out NULL, n_layers = 3 * cross layers <= DCTR_MLP_MAX_LAYERS.
 * A workgroup carries 16 samples through all layers (x_0, x_l and the rank-space tiles in LDS; fp32 MFMA 16x16x4).
 * The backward takes gY = d loss / d x_L and writes gx = d loss / d x_0 and, per dense layer, gW / gbias in the packed
 * form (the caller unpacks: gV, gG (summed over the cross layers), the diagonal blocks gC, gU, gb).
 * dctr_crossnet_mix_supported: W <= 512, E*R + E <= 512, E <= 8 (LDS).                                              */
int dctr_crossnet_mix_supported(int32_t W, int32_t n_cross_layers, int32_t E, int32_t R);
int dctr_crossnet_mix_fwd(const struct dctr_mlp* m, int32_t E, int32_t R, const float* x, int64_t ld_x, int32_t B,
                          dctr_stream_t stream);
size_t dctr_crossnet_mix_bwd_workspace_floats(const struct dctr_mlp* m, int32_t B);
int dctr_crossnet_mix_bwd(const struct dctr_mlp* m, int32_t E, int32_t R, const float* x, int64_t ld_x, int32_t B,
                          const float* gY, int64_t ld_g, float* gx, int64_t ld_gx, float* workspace,
                          dctr_stream_t stream);

/* ---- DNN tower + dnn_linear on fp32 MFMA (csrc/mlp.hip) ---------------------------------------------------
 * DNN.forward (layers/core.py:120-134) with relu (or linear) activations, no BatchNorm, dropout inactive:
 *     h_0 = x ;  h_{l+1} = act(h_l W_l^T + bias_l)            W_l = dnn.linears.<l>.weight [N_l, K_l]
 * optionally followed by the bias-free projection of deepfm.py:61,84 (xdeepfm / fibinet / pnn alike):
 *     logit[b] = h_L[b, :] . w_out                              w_out = dnn_linear.weight [1, N_L]
 * The struct lives in HOST memory.  W rows are addressed W + n*ld_w with ld_w % 4 == 0 and a 16-byte aligned
 * base; floats in the row padding [K, ld_w) must be finite (they are multiplied by zero).  gW is written with
 * the same leading dimension (padding columns receive 0).
 *   h   [B, ld_h] post-activation output of the layer: written by dctr_mlp_fwd when non-NULL (needed by the
 *                 backward for every layer; without w_out the last layer's h IS the result)
 *   dh  [B, ld_h] backward scratch: d loss / d pre-activation, written by dctr_mlp_bwd
 * dctr_mlp_bwd takes g = d loss / d logit [B] (with w_out) or d loss / d h_L [B, ld_g] (without) and writes
 *   gx [B, ld_gx] = d loss / d x (nullable), layer[l].gW, layer[l].gbias, g_w_out.
 * Replaces autograd's mm / addmm / threshold_backward / sum chains under basemodel.py:261.  Everything is
 * summed in a fixed order (no atomics): results are bit-reproducible.
 *   workspace  dctr_mlp_bwd_workspace_floats(m, B) floats (split-batch partials of the weight gradients)   */
#define DCTR_MLP_MAX_LAYERS 12
typedef struct dctr_mlp_layer {
  const float* W;
  const float* bias; /* [N] nullable                   */
  float* h;
  float* dh;
  float* gW;         /* [N, ld_w] nullable: not written */
  float* gbias;      /* [N] nullable                    */
  int32_t K, N;
  int32_t ld_w, ld_h;
  int32_t relu;      /* 1: relu, 0: identity            */
  int32_t pad_;
} dctr_mlp_layer_t;
typedef struct dctr_mlp {
  dctr_mlp_layer_t layer[DCTR_MLP_MAX_LAYERS];
  const float* w_out; /* [N_last] nullable */
  float* g_w_out;     /* [N_last] nullable */
  int32_t n_layers;
  int32_t pad_;
  int32_t* step_sync; /* nullable: dctr_mlp_train_step's first launch signals DCTR_SYNC_TOWER (gx and g_logit complete) */
} dctr_mlp_t;
size_t dctr_sizeof_mlp(void);
#ifdef DCTR_DIAG
/* diagnostics, DCTR_DIAG build only (tools/mlp_trace.py): buf = 3 x 4096 x 16 u64 of per-workgroup wall_clock64
 * stamps (forward | backward-data | wgrad); NULL switches tracing off.  Batches above 65536 samples are not traced
 * correctly.  */
void dctr_dbg_mlp_trace(unsigned long long* buf);
#endif
int dctr_mlp_fwd(const dctr_mlp_t* m, const float* x, int64_t ld_x, int32_t B, float* logit, dctr_stream_t stream);
size_t dctr_mlp_bwd_workspace_floats(const dctr_mlp_t* m, int32_t B);
int dctr_mlp_bwd(const dctr_mlp_t* m, const float* x, int64_t ld_x, int32_t B, const float* g, int64_t ld_g,
                 float* gx, int64_t ld_gx, float* workspace, dctr_stream_t stream);

/* The fused train step of a binary model whose logit is part0 + part1 + tower(x) + bias: tower forward, prediction
 * head, BCE(sum), backward-data (ONE launch per 16-sample row tile: the logits never leave the workgroup), then the
 * weight gradients and their reduction -- 3 launches for what dctr_mlp_fwd + dctr_bce_head + dctr_mlp_bwd do in 5.
 * Outputs: y_pred [B], loss [1], g_logit [B] (= d loss / d part0 = d loss / d part1), g_bias [1] (nullable), gx,
 * layer[l].gW / gbias, g_w_out.  Needs w_out and every layer's h / dh.
 * workspace: dctr_mlp_train_workspace_floats(m, B) floats.
 * defer_wgrad != 0: only the first launch is enqueued (y_pred, g_logit, gx, the h / dh activations are then
 * complete); the caller enqueues the weight-gradient half itself with dctr_mlp_train_wgrad (same m, x, workspace)
 * -- on another stream if it wants it to overlap with work that needs only gx / g_logit (the embedding update).
 * loss, g_bias, layer[l].gW / gbias and g_w_out are written by that second call.                               */
size_t dctr_mlp_train_workspace_floats(const dctr_mlp_t* m, int32_t B);
int dctr_mlp_train_step(const dctr_mlp_t* m, const float* x, int64_t ld_x, int32_t B, const float* part0,
                        const float* part1, const float* bias, const float* y, float* y_pred, float* loss,
                        float* g_logit, float* g_bias, float* gx, int64_t ld_gx, float* workspace,
                        int32_t defer_wgrad, const dctr_dense_step_t* step, dctr_stream_t stream);
/* step (nullable, both calls): the reduction that finishes layer[l].gW / gbias, g_w_out and g_bias also applies this
 * optimizer step to the parameters behind them -- no dctr_dense_opt launch, no join in front of it.               */
int dctr_mlp_train_wgrad(const dctr_mlp_t* m, const float* x, int64_t ld_x, int32_t B, const float* g_logit,
                         float* workspace, float* loss, float* g_bias, const dctr_dense_step_t* step,
                         dctr_stream_t stream);

/* ---- the embedding lookup as the fused train launch's input stage (round 4) ------------------------------------------
 * dctr_embed_fwd + dctr_mlp_train_step(defer_wgrad = 1) as ONE launch for a DeepFM / WDL-shaped model -- logit =
 * linear(X) [+ FM(embeddings)] + tower(combined_dnn_input) + bias (deepfm.py:67-86, wdl.py): every workgroup gathers the
 * table rows of its own 16 samples (basemodel.py:354-380), computes the linear logit (basemodel.py:63-92) and the FM
 * term (interaction.py:26-34) in LDS, and goes on with the tower, the head, BCE(sum) and the backward-data pass.  The
 * gather kernel, its output's round trip through memory and one kernel boundary leave the step's critical cycle.
 *   plan   fixed-length fields only, one embedding_dim in {4, 8, 16, 32, 64}, at most 32 wide fields; the tower's input is
 *          the plan's whole row (layer[0].K == n_deep * emb_dim + n_dense)
 *   out    [B, ld_out]  written: the gathered row of every sample (what dctr_embed_fwd's `out` holds): the operand of
 *          dctr_mlp_train_wgrad(m, out, ld_out, ...), which the caller enqueues as after dctr_mlp_train_step(defer_wgrad)
 *   fm_s   [B, ld_s]    written when non-NULL: sum_f e (FM's backward inside dctr_embed_update); required with want_fm
 *   y_pred, g_logit, gx, workspace, err: as in dctr_mlp_train_step / dctr_embed_fwd.
 * Same arithmetic, in the same order, as the two calls it replaces: results are bit-identical to theirs.
 * dctr_embed_tower_train_supported: 1 when plan + tower fit (else the caller keeps the two-launch path).            */
int dctr_embed_tower_train_supported(const dctr_plan_t* plan, const dctr_mlp_t* m, int32_t B);
int dctr_embed_tower_train_step(const dctr_plan_t* plan, const float* X, int64_t ldx, const dctr_mlp_t* m, int32_t B,
                                int32_t want_fm, const float* bias, const float* y, float* y_pred, float* g_logit,
                                float* gx, int64_t ld_gx, float* out, int64_t ld_out, float* fm_s, int64_t ld_s,
                                int32_t* err, float* workspace, dctr_stream_t stream);

/* The same two launches for a step whose critical cycle is tower -> embedding update -> next tower on ONE queue, with the
 * weight gradients on a second queue (round 6; reference: the optimizer step of basemodel.py:262 must precede the next
 * forward of basemodel.py:246 -- here that order is kept by a word in memory, see DCTR_SYNC_W_GEN below):
 *   dctr_mlp_train_wgrad_sync        dctr_mlp_train_wgrad as ONE launch: a tile's last workgroup to arrive sums the tile's
 *                                    partial slabs in slab order and steps the parameters (the arithmetic of the separate
 *                                    reduction launch, bit for bit), then the launch advances sync[DCTR_SYNC_W_GEN].
 *                                    counters: dctr_mlp_train_wgrad_counters(m, B) int32, zero before the first call.
 *   dctr_embed_tower_train_step_sync dctr_embed_tower_train_step that requests no dense parameter (tower weights, biases,
 *                                    the projection, `bias`) before sync[DCTR_SYNC_W_GEN] has caught up with the number of
 *                                    tower launches finished on this block; a wait that exceeds timeout_us raises bit 2 (4)
 *                                    of *err and goes on.  Results: those of the plain calls, bit for bit.                */
size_t dctr_mlp_train_wgrad_counters(const dctr_mlp_t* m, int32_t B);
int dctr_mlp_train_wgrad_sync(const dctr_mlp_t* m, const float* x, int64_t ld_x, int32_t B, const float* g_logit,
                              float* workspace, float* loss, float* g_bias, const dctr_dense_step_t* step, int32_t* sync,
                              int32_t* counters, dctr_stream_t stream);
int dctr_embed_tower_train_step_sync(const dctr_plan_t* plan, const float* X, int64_t ldx, const dctr_mlp_t* m, int32_t B,
                                     int32_t want_fm, const float* bias, const float* y, float* y_pred, float* g_logit,
                                     float* gx, int64_t ld_gx, float* out, int64_t ld_out, float* fm_s, int64_t ld_s,
                                     int32_t* err, float* workspace, int32_t* sync, int32_t timeout_us,
                                     dctr_stream_t stream);

/* ---- device-side dependencies between the two queues of a train step ---------------------------------------------
 * A dependency that crosses hardware queues costs 11-12 us through hipGraph / stream events on this stack, 4.6 us through
 * a word in memory (tools/micro/hopbench.hip), and the DeepFM step's critical cycle crosses twice (tower -> update,
 * gather -> tower: 24 of its 99 us).  So the producer kernels can SIGNAL and a one-wave kernel in front of the consumer
 * WAITS:
 *   sync block: DCTR_SYNC_INTS int32, zero before first use, one per model.  For each signal s (DCTR_SYNC_TOWER /
 *     DCTR_SYNC_GATHER): a generation counter the producer's LAST workgroup advances -- after every workgroup has
 *     stored what the consumer reads with write-through stores and waited for them -- and an epoch counter the waiter
 *     advances once per call: the n-th wait returns when the n-th signal has been given.
 *   dctr_step_wait enqueues the waiter (1 workgroup of 64 threads: it cannot keep a producer from being scheduled).  It
 *     gives up after timeout_us (and raises bit s of the block's error word, DCTR_SYNC_ERR): a kernel stream that is
 *     serialised by a profiler collecting counters, or a producer that was never launched, costs time, not a hang.
 * The consumer is launched behind the waiter on the same stream; its own start-of-kernel acquire does the rest.
 * Signals and waits must pair up one to one (the fused train step does; dctr_step_sync is for nothing else).        */
#define DCTR_SYNC_TOWER 0
#define DCTR_SYNC_GATHER 1
#define DCTR_SYNC_UPDATE 2   /* given by dctr_step_signal behind dctr_embed_update (the "fused_flags" step topology) */
#define DCTR_SYNC_ERR 12     /* index of the error word */
#define DCTR_SYNC_INTS 32    /* [4 s, 4 s + 4): signal s' generation / epoch / arrivals / stamp; [16 + 2 s, +2): its waiter's stamps */
/* Round 6 -- the weights' hand-over INSIDE the waiting kernel (no waiter launch, no graph edge): words of the same block.
 *   W_GEN  weight steps finished: advanced by the last reducer of dctr_mlp_train_wgrad_sync, after every dense parameter
 *          it steps has been stored write-through and waited for;
 *   T_GEN  tower launches finished: advanced by the last workgroup of dctr_embed_tower_train_step_sync.
 * A tower launch may start while the previous step's weight-gradient launch still runs on another queue: it stages its X
 * tile and gathers its table rows (which only need the embedding update, ordered in front of it on its own queue), then
 * waits until W_GEN >= T_GEN (as read at its start) before it requests the first dense parameter.  The two calls must
 * alternate strictly (tower, weight gradients, tower, ...) on one sync block; the block is zero before the first call.  */
#define DCTR_SYNC_W_GEN 24
#define DCTR_SYNC_W_ARR 25
#define DCTR_SYNC_T_GEN 26
#define DCTR_SYNC_T_ARR 27
int dctr_step_wait(int32_t* sync, int32_t signal, int32_t timeout_us, dctr_stream_t stream);
/* The signal as a one-thread launch of its own on the producer's queue, behind the producer (whose end-of-kernel
 * write-back makes its stores visible first): for a producer that cannot signal from inside its kernel.            */
int dctr_step_signal(int32_t* sync, int32_t signal, dctr_stream_t stream);
/* *dst (device) = the 100 MHz device wall clock at the moment a one-thread launch runs on `stream`: put between two kernels
 * of a queue it dates their boundary -- the only way to time a kernel INSIDE a hipGraph replay (bench.py: the graph-replayed
 * duration of the step's dominant kernels).                                                                           */
int dctr_stamp(uint64_t* dst, dctr_stream_t stream);

/* ---- direct exchange between the ranks of a table-sharded job (one process per GPU; deepctr_torch/parallel.py) ----------
 * Replaces the host-issued RCCL all-to-alls / all-reduce of the sharded step (basemodel.py:206-209 is nn.DataParallel in
 * the reference) by copies into the peers' IPC-mapped receive buffers + a word per (exchange, sender): everything is an
 * ordinary stream operation, so the WHOLE step -- exchanges included -- is one hipGraph.  `step`: this rank's exchange
 * counter (device int32); peer_words[r]: rank r's word array for this exchange (device pointers, IPC-mapped), my_index:
 * this rank's slot in it.  Post behind the copies (same stream); wait in front of the consumer; next advances the counter.
 * A wait gives up after timeout_us and raises bit 1 of *err.  dctr_sum_ranks: dst[i] = sum_r src[r * ld + i] in rank
 * order (the dense gradients' all-reduce = all-gather by copy + this sum: every rank lands on the same bits).          */
/* hipMemcpyAsync(device to device) on the caller's stream (dst may be another device's memory mapped through IPC);
 * dctr_enable_peer_access(d): the current device may reach device d's memory (no-op for d == current device,
 * DCTR_ENOSUP when the devices have no peer path).                                                                  */
int dctr_copy_async(void* dst, const void* src, size_t bytes, dctr_stream_t stream);
int dctr_enable_peer_access(int32_t peer_device);
int dctr_exchange_post(int32_t* const* peer_words, int32_t n, int32_t my_index, const int32_t* step, dctr_stream_t stream);
int dctr_exchange_wait(const int32_t* words, int32_t n, const int32_t* step, int32_t timeout_us, int32_t* err,
                       dctr_stream_t stream);
int dctr_exchange_next(int32_t* step, dctr_stream_t stream);
int dctr_sum_ranks(float* dst, const float* src, const uint64_t* src_tbl, int32_t n_ranks, int64_t n, int64_t ld,
                   const dctr_dense_step_t* step, int32_t store, dctr_stream_t stream);
/* step (nullable): the optimizer step on the elements it finishes.  src_tbl [n_ranks] (device, nullable): rank r's slab
 * is read where it lies, at (const float*)src_tbl[r] (16-byte aligned, peer memory) -- a pull, no copies; store == 0
 * (needs step): the sum itself is not written, dst only names the gradients' slab positions and may alias a source.  */
/* post + wait (+ advance != 0: the counter's advance, behind a step's last exchange) as one launch */
int dctr_exchange_sync(int32_t* const* peer_words, const int32_t* words, int32_t n, int32_t my_index, int32_t* step,
                       int32_t advance, int32_t timeout_us, int32_t* err, dctr_stream_t stream);
/* One launch in front of a sharded step: this rank's batch (xb [B, ncols], yb [B]) into the static buffers the captured
 * step reads, and -- x_next non-NULL -- the next batch's id columns id_cols [n_ranks * n_slots] into columns
 * [ids_col, ids_col + n_slots) of the gradient chunks send [n_ranks][B][ld_chunk] (they ride in the gradient exchange). */
int dctr_shard_stage(const float* xb, int64_t ld_xb, const float* yb, int32_t B, int32_t ncols, float* x_dst,
                     int64_t ld_xd, float* y_dst, const float* x_next, int64_t ld_xn, const int32_t* id_cols,
                     int32_t n_ranks, int32_t n_slots, float* send, int64_t ld_chunk, int32_t ids_col,
                     dctr_stream_t stream);

/* ---- prediction head + loss (layers/core.py:154-160, basemodel.py:254, F.binary_cross_entropy(reduction='sum'))
 *     z = sum_i part_i[b] + bias ;  y_pred = sigmoid(z) ;  loss = sum_b -(y log p + (1-y) log(1-p))   (logs clamped
 *     at -100 like ATen) ;  g_logit = d loss / d z as autograd computes it:
 *     (p - y) / max((1-p) p, 1e-12) * (1-p) p ;  g_bias = sum_b g_logit.
 * Up to four logit parts (linear, FM, DNN, CIN ...), each [B], nullable.  One workgroup, tree reductions:
 * deterministic.  Replaces ~12 elementwise / reduce launches per step.                                      */
int dctr_bce_head(const float* part0, const float* part1, const float* part2, const float* part3, const float* bias,
                  const float* y, int32_t B, float* y_pred, float* loss, float* g_logit, float* g_bias,
                  dctr_stream_t stream);

/* ---- autograd glue as single launches (models on torch.autograd: xDeepFM, FiBiNET, DCN, PNN, ...) ------------------------
 * dctr_rows_join: out[b, 0:W) = a[b, :] (+ c[b, :]), out[b, W:W+n_d) = d[b, :], out[b, W+n_d:ld_out) = 0 -- the gradient of
 * the gather's output [B, ld_out] from the gradients of its two views (replaces the slice backward's copies + fill, and
 * autograd's add when c is given; a, c, d nullable).  W, ld_* multiples of 4, 16-byte aligned rows.  out may BE a (same
 * leading dimension): every lane reads the words it writes -- "a[:, :W] += c, a[:, W:] = 0" in one launch.
 * dctr_relu_bwd_bias: g_out = g * (h > 0) (aten::threshold_backward; h NULL: g_out = g; g_out NULL: not written) and
 * g_bias[n] = sum_b g_out[b, n] in a fixed order (replaces threshold_backward + sum(0) behind a wide nn.Linear,
 * layers/core.py:120-134).  workspace: dctr_relu_bwd_bias_workspace_floats(B, N) floats.                             */
int dctr_rows_join(const float* a, int64_t ld_a, const float* c, int64_t ld_c, int32_t W, const float* d, int64_t ld_d,
                   int32_t n_d, float* out, int64_t ld_out, int32_t B, dctr_stream_t stream);
size_t dctr_relu_bwd_bias_workspace_floats(int32_t B, int32_t N);
int dctr_relu_bwd_bias(const float* g, int64_t ld_g, const float* h, int64_t ld_h, int32_t B, int32_t N, float* g_out,
                       int64_t ld_o, float* g_bias, float* workspace, dctr_stream_t stream);

/* ---- dense optimizer over one flat parameter slab (torch.optim.SGD / Adagrad, basemodel.py:447-461) --------
 *   DCTR_UPD_SGD      p -= lr * g
 *   DCTR_UPD_ADAGRAD  state += g*g ; p -= lr * g / (sqrt(state) + eps)
 * One launch for every dense parameter of the model (the reference's optimizer issues ~8 foreach launches).  */
int dctr_dense_opt(float* p, const float* g, float* state, int64_t n, int32_t opt, float lr, float eps,
                   dctr_stream_t stream);
/* The same step on a list of contiguous fp32 tensors -- the parameters of a model whose train step runs through
 * autograd (xDeepFM's CIN weights, DCN's cross kernels, ...), each with the gradient autograd left in `.grad` and its
 * Adagrad `sum` -- in one launch per 48 tensors (torch.optim.Adagrad / SGD step, basemodel.py:262: 5 foreach launches).
 * items: HOST array.  state: ignored for DCTR_UPD_SGD.                                                              */
typedef struct dctr_dense_item {
  float* p;
  const float* g;
  float* state;
  int64_t n;
  float l2;   /* lambda of an L2 term lambda * sum(p^2) on this tensor (basemodel.py:412-428), 0: none.  The step then
               * uses g + 2*lambda*p -- what autograd adds to .grad for the term (rounded like autograd: product, sum) */
  float pad_;
} dctr_dense_item_t;
size_t dctr_sizeof_dense_item(void);
int dctr_dense_opt_multi(const dctr_dense_item_t* items, int32_t n_items, int32_t opt, float lr, float eps,
                         dctr_stream_t stream);
/* out[0] = sum_i l2_i * sum(p_i^2): the value of those terms for the logged loss; one workgroup, fixed summation order.
 * (g / state of the items are not read.)                                                                            */
int dctr_l2_value_multi(const dctr_dense_item_t* items, int32_t n_items, float* out, dctr_stream_t stream);

/* ---- table-sharded multi-GPU exchange: the two "assemble" kernels (csrc/shard.hip, deepctr_torch/parallel.py) ---
 * Rank q of N owns units q, q+N, q+2N, ... (a unit = one id column with its deep and/or wide table).  Owners gather
 * with dctr_embed_fwd over the N*B global samples and update with dctr_embed_update; what travels between ranks
 * are chunks [B, ld_chunk] whose row b is [ slot 0 | slot 1 | ... (slot j = unit q + j*N, D floats each) | pad |
 * wide partial sum at column wide_col | pad ].
 * dctr_shard_assemble_fwd: recv [N][B][ld_chunk] (chunk q from owner q) -> the single-GPU outputs of
 *   dctr_embed_fwd: out [B, ld_out] = [ e_0 | ... | e_{F-1} | dense block at dense_off ], wide [B] (sum of the owners'
 *   partials in rank order + X[:, wdense_cols] . wdense_w), fm [B], fm_s [B, ld_s]  (wide / fm / fm_s nullable).
 * dctr_shard_assemble_bwd: the adjoint; send [N][B][ld_chunk] with G[b, f] = g_out[b, f] + g_fm[b] * (S[b] - e[b, f])
 *   in unit f's slot and g_wide[b] at wide_col; g_wdense [n_wdense] (nullable) = X[:, wdense_cols]^T g_wide.
 * owner_slot [F] (device, nullable): unit f is owned by rank (owner_slot[f] & 0xffff) and sits in slot
 *   (owner_slot[f] >> 16) of that owner's chunk; NULL: owner f % N, slot f / N.
 * send_chunks [N] (device, nullable): chunk q is written at (float*)send_chunks[q] ([B][ld_chunk]) instead of
 *   send + q * B * ld_chunk -- owner q's receive buffer, peer memory: the push-style exchange, no copy node; columns
 *   [carry_col, carry_col + carry_n) of the LOCAL send rows (dctr_shard_stage's next-batch ids) are copied along.
 * Deterministic (no atomics).  D <= 64.                                                                          */
int dctr_shard_assemble_fwd(const float* recv, int64_t ld_chunk, int32_t n_ranks, int32_t B, int32_t F, int32_t D,
                            const int32_t* owner_slot, int32_t wide_col, const float* X, int64_t ld_x, const int32_t* dense_cols,
                            int32_t n_dense, int32_t dense_off, const int32_t* wdense_cols, const float* wdense_w,
                            int32_t n_wdense, float* out, int64_t ld_out, float* wide, float* fm, float* fm_s,
                            int64_t ld_s, dctr_stream_t stream);
int dctr_shard_assemble_bwd(float* send, const uint64_t* send_chunks, int32_t carry_col, int32_t carry_n, int64_t ld_chunk, int32_t n_ranks, int32_t B, int32_t F, int32_t D,
                            const int32_t* owner_slot, int32_t wide_col, const float* g_out, int64_t ld_g, const float* g_wide, const float* g_fm,
                            const float* out, int64_t ld_out, const float* fm_s, int64_t ld_s, const float* X,
                            int64_t ld_x, const int32_t* wdense_cols, int32_t n_wdense, float* g_wdense,
                            dctr_stream_t stream);
/* The same with the carried columns taken straight from the NEXT batch's input matrix (round 6: the staging launch in front
 * of every step of a captured group is gone): carried value (q, j) of sample b = x_next[b, carry_cols[q * carry_n + j]]
 * (carry_cols [n_ranks * carry_n], device); x_next == NULL: dctr_shard_assemble_bwd.                                    */
int dctr_shard_assemble_bwd_next(float* send, const uint64_t* send_chunks, int32_t carry_col, int32_t carry_n, int64_t ld_chunk,
                                 int32_t n_ranks, int32_t B, int32_t F, int32_t D, const int32_t* owner_slot, int32_t wide_col,
                                 const float* g_out, int64_t ld_g, const float* g_wide, const float* g_fm, const float* out,
                                 int64_t ld_out, const float* fm_s, int64_t ld_s, const float* X, int64_t ld_x,
                                 const int32_t* wdense_cols, int32_t n_wdense, float* g_wdense, const float* x_next,
                                 int64_t ld_xn, const int32_t* carry_cols, dctr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DCTR_H */
