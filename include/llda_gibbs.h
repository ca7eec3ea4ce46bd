/* llda_gibbs.h -- C ABI of the MI355X-native collapsed-Gibbs sweep for Labeled LDA / CascadeLDA.
 *
 * The reference (KenHBS/LDA_thesis) is pure Python and has no FFI boundary of its own; the hot path
 * is the method  LabeledLDA.training_iteration  (/root/reference/LabeledLDA.py:101-125), byte-for-byte
 * the same body as  SubLDA.training_iteration  (/root/reference/CascadeLDA.py:397-421).  This header
 * is the boundary a maintainer binds (ctypes stub in INTEGRATION.md) to replace those two methods,
 * the count initialisation loops (LabeledLDA.py:89-92, CascadeLDA.py:382-385) and the thinning
 * read-outs (LabeledLDA.py:231-239, :256-265).
 *
 * Conventions
 *   - every pointer marked [dev] is a DEVICE pointer (HBM of the current HIP device); the library
 *     allocates nothing persistent and owns none of the buffers;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all entry points only
 *     ENQUEUE work and return without synchronising unless stated otherwise;
 *   - return value 0 = success, negative = LLDA_E_* (see llda_strerror); HIP failures are returned
 *     as LLDA_E_HIP and the hipError_t is available from llda_last_hip_error(); nothing throws;
 *   - one host thread per device; entry points are re-entrant across devices.
 *
 * Device layout ("group layout")
 *   The K topics of a document are spread over G lanes x T slots; per-topic arrays are stored in
 *   device order with padded row length KP = G*T.  The position of (lane, slot) in a row is
 *       pos = ((slot / 4) * G + lane) * 4 + slot % 4      when T is a multiple of 4
 *       pos = lane * T + slot                               when T is 1 or 2
 *   i.e. the 16-byte chunk `slot / 4` of ALL lanes is contiguous, so that every vector load of a wavefront reads
 *   one contiguous run of the row (llda_layout.pos_lane / pos_slot / topic_pos hold the permutation):
 *       n_kw  [V][KP] int32   word-major  (the reference's n_k_v (K,V) transposed + permuted)
 *       n_dk  [D][KP] int32   (the reference's n_d_k (D,K) permuted)
 *       n_k   [KP]    int32   (the reference's n_zk permuted)
 *       z     [S]     int32   topic of every site, stored as device POSITION (not topic id)
 *       lab_mask [D][G] uint16   bit s of [d][g] = labs[d][topic at (g,s)]
 *   llda_layout() returns the permutation so the host can convert to and from reference order.
 */
#ifndef LLDA_GIBBS_H
#define LLDA_GIBBS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLDA_ABI_VERSION 21
#define LLDA_MAX_K 7688          /* every K up to here splits into <= 64 pairwise leaves                       */
#define LLDA_MAX_KP 8192         /* longest padded row: 64 leaves x 128                                        */
#define LLDA_MAX_LEAVES 8        /* "narrow" layouts: one lane group of <= 64 lanes x <= 16 slots per document  */
#define LLDA_MAX_WIDE_LEAVES 64  /* "wide" layouts (more than 8 leaves: some K in 969..1023, every K > 1024)    */
#define LLDA_MAX_ROUNDS 4

enum {
    LLDA_OK = 0,
    LLDA_E_BAD_K = -1,      /* K outside 1..LLDA_MAX_K, or an entry point that only takes narrow layouts */
    LLDA_E_BAD_ARG = -2,    /* NULL pointer, negative size, layout fields do not match K   */
    LLDA_E_HIP = -3,        /* a HIP runtime call failed (see llda_last_hip_error)          */
    LLDA_E_NO_DEVICE = -4   /* no gfx950 device visible                                     */
};

/* Layout of K topics over the lanes of a wavefront (host-side description).
 * Narrow layouts (n_leaves <= 8): G = 8 * (leaves rounded up to a power of two) <= 64 lanes per document, 64/G documents
 * per wavefront.  Wide layouts (wide = 1; ABI 13): G = 8 * (leaves rounded up to a multiple of 8) = 64 * tiers; the G
 * "lanes" of the formulas above are VIRTUAL lanes, virtual lane gv = 64 * tier + physical lane of the one wavefront that
 * walks the document; T is 8, 12 or 16.  The leaf totals of a wide layout are combined along numpy's recursion tree by
 * n_leaves - 1 in-place adds  total[comb_dst[i]] += total[comb_src[i]]  (post-order; total[0] is the sum).  Every entry
 * point takes wide layouts except llda_sweep_batch (its problems have at most 128 topics). */
typedef struct llda_layout {
    int32_t K;                       /* number of topics (labels incl. 'root')                */
    int32_t n_leaves;                /* numpy pairwise-sum leaves                             */
    int32_t G;                       /* lanes per document: 8, 16, 32, 64; wide: 64 * tiers   */
    int32_t T;                       /* slots per lane: 1, 2, 4, 8, 12 or 16                  */
    int32_t KP;                      /* G*T                                                   */
    int32_t tail;                    /* K % 8 of the last leaf                                */
    int32_t tail_row;                /* slot that holds the tail topics                       */
    int32_t n_rounds;                /* leaf-combine rounds (narrow)                          */
    int32_t wide;                    /* 1: more than 8 leaves                                 */
    int32_t tiers;                   /* wide: G / 64; narrow: 0                               */
    int32_t leaf_start[LLDA_MAX_WIDE_LEAVES];
    int32_t leaf_len[LLDA_MAX_WIDE_LEAVES];
    int32_t rounds[LLDA_MAX_ROUNDS][LLDA_MAX_LEAVES];   /* narrow: partner leaf per round (self = idle) */
    int32_t comb_dst[LLDA_MAX_WIDE_LEAVES];             /* numpy's recursion tree, post-order (n_leaves - 1 entries) */
    int32_t comb_src[LLDA_MAX_WIDE_LEAVES];
    int32_t topic_pos[LLDA_MAX_KP];  /* topic id -> device position                           */
    int32_t pos_topic[LLDA_MAX_KP];  /* device position -> topic id, -1 in the padding        */
    int32_t pos_lane[LLDA_MAX_KP];   /* device position -> (virtual) lane that holds it        */
    int32_t pos_slot[LLDA_MAX_KP];   /* device position -> slot of that lane (bit of lab_mask) */
} llda_layout;

/* Arguments of one sweep over a shard of documents.
 * Replaces the loop body of LabeledLDA.training_iteration (LabeledLDA.py:106-125) for documents
 * [doc_base, doc_base + D) under per-document snapshot semantics: every document reads the
 * sweep-start n_kw / n_k plus its own changes; count changes are accumulated into the *_delta
 * buffers with integer atomics (order independent => deterministic). */
typedef struct llda_sweep_args {
    const int64_t  *doc_off;     /* [dev] [D+1] site offsets                                  */
    const int32_t  *doc_order;   /* [dev] [D] processing order (local doc ids) or NULL        */
    const int32_t  *word;        /* [dev] [S] word id of every site (unique, ascending per doc) */
    const int32_t  *freq;        /* [dev] [S] frequency f of every site, 0 <= f < 2^23 (not checked on the device) */
    int32_t        *z;           /* [dev] [S] in/out: device position of the site's topic      */
    const uint16_t *lab_mask;    /* [dev] [D*G] lane masks                                     */
    int32_t        *n_dk;        /* [dev] [D*KP] in/out                                        */
    const int32_t  *n_kw;        /* [dev] [V*KP] sweep-start snapshot (read only)              */
    int32_t        *n_kw_delta;  /* [dev] [V*KP] += sweep changes                              */
    const int32_t  *n_k;         /* [dev] [KP] sweep-start snapshot (read only)                */
    int32_t        *n_k_delta;   /* [dev] [KP] += sweep changes                                */
    int32_t        *status;      /* [dev] optional (may be NULL), int32[4]: word 0 bit 0 is set when a site had no
                                    topic with positive probability (the reference would raise);
                                    bit 1 (informational) when some site took the exact tier;
                                    bit 2 when llda_pack_rows16 met a count outside 0 .. 65535 in a flagged row, or the
                                    four-wave 16-bit-row kernel an entry of n_dk above 65535 (max_doc_tokens was not a bound);
                                    word 1 += sites the fp32 tier was unsure about, word 2 += sites that
                                    reached the exact tier (statistics)                           */
    int64_t  D;                  /* local documents                                            */
    int64_t  V;                  /* vocabulary size (rows of n_kw; also enters den = n_k + V*beta) */
    int32_t  K;                  /* topics                                                     */
    int32_t  docs_per_group;     /* documents a lane group walks per workgroup (>=1; 0 = auto) */
    int32_t  dense_mask;         /* 1 = lab_mask allows every topic in every document (the kernel may
                                    then skip applying it); 0 = general                            */
    int32_t  debug_margin;       /* 0 in production.  Test hook of the two-tier draw: n > 0 widens the
                                    tier-1 safety margin to 2^-n of the total score (more sites take the
                                    exact tier), -1 sends every site through the exact tier, -2 skips only
                                    the fp32 tier 0; -3 (wide layouts only) production margins on the kernel
                                    that keeps nothing of the row in registers, -4 on the register kernel with
                                    LDS copies of the counts whatever max_doc_tokens says, -5 on the fp64 register
                                    kernel without its fp32 tier, -6 the fp32 tier with fp64 factors in LDS even when
                                    scratch is there, -7 with fp32 factors only whenever scratch is there; -8 (with n_kw16)
                                    production margins on the three-wave form of the 16-bit-row kernel; -9 (with row16) the quad
                                    kernel with the constant tier-0 margin 104 * 2^-24 of the total instead of its data-dependent one;
                                    -10 ... -18 (with row16) the data-dependent margin scaled by 1 / 1.05 (the derived error bound itself),
                                    1/2, 1/4 ... 1/256: tests/neartie.py measures how much of the margin the worst planted tie needs */
    double   alpha, beta;        /* priors (LabeledLDA.py:55-56)                               */
    uint64_t seed;               /* RNG key                                                    */
    uint32_t sweep;              /* RNG counter word 3                                         */
    uint32_t stream_id;          /* RNG counter word 2 (sub-problem id for CascadeLDA)         */
    int64_t  doc_base;           /* global id of local document 0 (RNG counter word 1)         */
    /* optional sparse-label path (live_off, live_pos non-NULL and live_max <= 64): one lane per allowed
     * topic instead of one lane per 16 topics */
    const int64_t *live_off;     /* [dev] [D+1] offsets into live_pos                              */
    const int32_t *live_pos;     /* [dev] device positions of the topics every document allows, in DRAW order:
                                  * ascending (lane, slot) of the layout -- llda_layout.pos_lane / pos_slot --,
                                  * which for layouts with T >= 8 is not ascending memory position          */
    void          *scratch;      /* [dev] optional (ABI 14) work space: llda_sweep_scratch_bytes(K, D) bytes, contents irrelevant.
                                    Wide layouts of three or four tiers (K = 1 929 ... 3 848) with a dense or general label mask
                                    keep fp32 factors only in LDS when it is there (more wavefronts per CU) and run the rare
                                    fp64 / exact tiers on it.  NULL or too small: the same result from the kernel that keeps
                                    fp64 factors in LDS (14 % slower there)                                                 */
    int64_t  scratch_bytes;      /* size of scratch                                                        */
    int32_t  live_max;           /* largest number of allowed topics of any document                  */
    int32_t  max_doc_tokens;     /* (ABI 13) optional hint, 0 = unknown: an upper bound of every entry of n_dk of the call --
                                    the tokens (sum of freq, plus any phantom counts) of its largest document.  Wide layouts:
                                    below 32 768 the kernel keeps the document's count CHANGES as int16 in LDS instead of copies
                                    of the counts (more wavefronts per CU).  With n_kw16: below 65 536 the kernel packs n_dk with
                                    its sweep-start value into one LDS word and runs FOUR waves per SIMD instead of three
                                    (debug_margin -8: three regardless); an entry above the bound sets status bit 2         */
    /* optional commit log (both non-NULL): instead of two int32 atomics on n_kw_delta per changed site the
     * kernels store ONE word (old position | new position << 16) at the site's place in WORD-major order;
     * llda_commit_log then folds the log into the counts word by word without global atomics.  (No-return
     * global atomics saturate at ~27 G/s on MI355X, scattered 4-byte stores at ~88 G/s: tools/atomic_ubench.hip.)
     * n_kw_delta is not touched in this mode. */
    const int32_t *csc_pos;      /* [dev] [S] index of every site in word-major (stable) order        */
    uint32_t      *commit_log;   /* [dev] [S] out, word-major                                         */
    int64_t  n_sites;            /* doc_off[D] - doc_off[0], the sites this call spans: must be < 2^30 (the hot
                                    kernel addresses word / freq / z / csc_pos as base + 32-bit byte offset from
                                    the first document of the call).  Larger shards: one call per document
                                    range -- pass doc_off + d0, D = d1 - d0, n_dk + d0*KP, lab_mask + d0*G,
                                    doc_base + d0 (and live_off + d0); every other pointer stays as it is. */
    const int32_t *site_rec;     /* [dev] [S][4] optional (ABI 12), with the commit log: {word, freq, csc_pos, 0} of
                                    every site as one 16-byte record.  Layouts with 8 or 16 lanes per document (K <= 256)
                                    read it instead of the three arrays: a wavefront then walks 8 / 4 documents and
                                    every scalar load touches that many cache lines, which makes the vector-memory
                                    address pipeline the bound.  A call that passes it with such a layout may span at
                                    most 2^28 - 1 sites. */
    const uint16_t *n_kw16;      /* [dev] [V*KP] optional (ABI 15): the 16-bit image of n_kw written by llda_pack_rows16 for THIS
                                    sweep's n_kw.  Sites whose csc_pos has bit 31 set read their word's row from it (half the
                                    bytes for the rows that miss the L2); the caller sets the bit for the sites of flagged
                                    words only.  Taken by the dense 16-slot kernel with the commit log (llda_rows16_ok(K),
                                    dense_mask = 1, alpha, beta >= 1e-6); LLDA_E_BAD_ARG on any other path.  Results do
                                    not depend on it. */
    const int32_t *site_row;     /* [dev] [S] with n_kw16 (ABI 16; LLDA_E_BAD_ARG when one comes without the other): where the
                                    row of a site's word starts, in 16-byte units from n_kw --  word * KP / 4  for an int32 row,
                                    ((char *)n_kw16 - (char *)n_kw) / 16 + word * KP / 8  for a 16-bit one (int32 offsets: the caller
                                    carves both arrays out of ONE allocation, so the distance is a constant of V and KP).  Static: which words are flagged never changes.  The kernel reads it
                                    INSTEAD of word -- the choice between the two images costs it no instruction. */
    const void    *n_kw_img;     /* [dev] [V*KP] optional (ABI 18), sparse label sets only (live_off / live_pos; LLDA_E_BAD_ARG on
                                    any other path): the narrow image of THIS sweep's n_kw written by llda_pack_image -- one
                                    byte (img_bits 8) or one 16-bit word (img_bits 16) per count, SATURATING at 255 / 65535.
                                    The sparse-label kernel gathers its counts from the image (a row spans a quarter / half
                                    as many cache lines; that kernel is bound by L2 line fills) and re-reads an entry that
                                    shows the saturation value from n_kw itself, so any count is legal and the results do not
                                    depend on the image.  Needs alpha, beta >= 1e-6 like the sparse-label kernel itself. */
    int32_t  img_bits;           /* 0 (no image), 8 or 16                                                    */
    int32_t  reserved_img;       /* 0                                                                         */
    const uint8_t *row16;        /* [dev] [V] optional (ABI 19), with n_kw16 and WITHOUT site_row: the per-word flags llda_pack_rows16_all
                                    wrote for THIS sweep's n_kw (1 = every count of the row fits 16 bits; n_kw16 then holds EVERY
                                    row).  K with llda_quad_ok (512; ABI 21: the layouts of 16 slots per lane with 8, 16 or 32 lanes), dense_mask = 1, the commit log,
                                    alpha, beta >= 1e-6, 0 < max_doc_tokens < 65 536 and, for K = 128 and 256, site_rec (LLDA_E_BAD_ARG otherwise): the kernel that
                                    walks a document with K / 32 lanes x 32 slots (kernel_quad.hpp), FOUR documents per wavefront at
                                    K = 512, eight at 256, sixteen at 128 -- the per-iteration work of scan, search, pick and count
                                    update is shared by that many sites.  A site whose row is not flagged
                                    reads the int32 row (no prefetch: meant to be rare).  Bit 31 of csc_pos is ignored.  Results do
                                    not depend on it. */
    const int32_t *img_col;      /* [dev] [KP] optional (ABI 20), with n_kw_img: the image column that holds the count of every device
                                    position -- the inverse of the col_src handed to llda_pack_image_cols.  The sparse-label kernel is
                                    bound by the L2's line fills: an image whose columns are ordered so that topics which are allowed
                                    TOGETHER sit in one 128-byte line costs a site fewer fills.  NULL: column = position
                                    (llda_pack_image).  Results do not depend on it. */
} llda_sweep_args;

/* ---- host-only (no device needed) ---- */
int         llda_abi_version(void);
/* 0 for the production build.  Otherwise one bit per compile-time switch the library was built with (ABI 17): overrides of the
 * draw margins / occupancy bound, and the ablation switches of tools/ that delete work from the kernels.  A number measured
 * with a non-zero build is not a measurement of the product: bench.py refuses to print one. */
#define LLDA_BUILD_MARGIN0            0x001   /* -DLLDA_MARGIN0=...       tier-0 margin of the narrow kernels            */
#define LLDA_BUILD_WAVES              0x002   /* -DLLDA_WAVES=...         waves per SIMD the narrow kernels are bounded to */
#define LLDA_BUILD_MARGIN0_WIDE       0x004   /* -DLLDA_MARGIN0_WIDE=...                                                */
#define LLDA_BUILD_ABL_NOLOAD         0x008   /* -DABL_NOLOAD             no n_kw row loads                               */
#define LLDA_BUILD_ABL_NOCOMMIT       0x010   /* -DABL_NOCOMMIT           no count updates                                */
#define LLDA_BUILD_ABL_WIDE_NOROW     0x020   /* -DABL_WIDE_NOROW                                                         */
#define LLDA_BUILD_ABL_WIDE_NOADDLOAD 0x040   /* -DABL_WIDE_NOADDLOAD                                                     */
#define LLDA_BUILD_ABL_NOFMA          0x080   /* -DABL_NOFMA                                                              */
#define LLDA_BUILD_ABL_EXTRA_LDS      0x100   /* -DABL_EXTRA_LDS_BYTES=n  unused dynamic LDS (occupancy ablation)          */
#define LLDA_BUILD_QUAD_PROFILE       0x200   /* -DQUAD_PROFILE           shader-clock stamps in the quad kernel (status[8..]) */
#define LLDA_BUILD_BUDGET_MARKS       0x400   /* -DLLDA_BUDGET_MARKS      class marks in the assembly of the site loops (counting only) */
#define LLDA_BUILD_QUAD_PRIO          0x800   /* -DLLDA_QUAD_PRIO=tdb     issue priorities of the quad kernels' phases (experiments)    */
int         llda_build_info(void);
const char *llda_strerror(int code);
int         llda_last_hip_error(void);
/* sizeof of the argument structs as the library was compiled (0 llda_layout, 1 llda_sweep_args, 2 llda_batch_args,
 * 3 llda_foldin_args): a binding checks its own struct definitions against it once, at load time. */
int         llda_struct_size(int which);
/* Fill *out for K topics.  Mirrors numpy's pairwise-sum recursion (np.sum at LabeledLDA.py:117). */
int         llda_layout_init(int32_t K, llda_layout *out);

/* Bytes of llda_sweep_args.scratch that let llda_sweep(K, D documents per call) run its fastest kernel (0 for layouts that
 * need none: every narrow layout).  Host only. */
int64_t     llda_sweep_scratch_bytes(int32_t K, int64_t D);
/* 1 when llda_sweep can read 16-bit rows for K topics (narrow layout, 16 slots per lane, 32 or 64 lanes, no padded slot:
 * K = 512 and K = 1024).  Host only. */
int         llda_rows16_ok(int32_t K);
/* 1 when llda_sweep takes llda_sweep_args.row16 for K topics (ABI 21; narrow layout of 16 slots per lane whose 8, 16 or 32 lanes all
 * belong to a leaf of numpy's pairwise sum: K = 97 .. 128 (one leaf), 185 .. 248 without 192 and 256 (two), 361 .. 488 without 368,
 * 376, 384 and 496, 504, 512 (four) -- K = 100, 120, 200, 240, 400, 480 as well as 128, 256, 512; not 250 (three leaves in a four-leaf
 * layout) or 500 (five leaves).  Positions without a topic
 * keep the factor 0, the exact tier sums in numpy's order for K.  Host only. */
int         llda_quad_ok(int32_t K);

/* ---- device entry points (enqueue on `stream`) ---- */
/* One Gibbs sweep over the shard: LabeledLDA.py:101-125 / CascadeLDA.py:397-421. */
int llda_sweep(const llda_sweep_args *args, void *stream);

/* One Gibbs sweep over MANY independent small problems in ONE launch: the ensemble of per-node sub-problems of
 * CascadeLDA.go_down_tree (/root/reference/CascadeLDA.py:135-184; each is SubLDA.training_iteration,
 * CascadeLDA.py:397-421).  A "document instance" is a document as a member of one sub-problem.  All problems
 * share the vocabulary (V) and the priors; problem p keeps its fused [n_kw (V x KP_p) | n_k (KP_p)] counts at
 * counts + kw_off[p] / counts + nk_off[p], and its changes are added (int32 atomics) at the same offsets of
 * `delta`; the caller folds with llda_apply_delta(counts, delta, total) after the launch(es) of a sweep.
 * The draw key is (seed; sweep, stream = prob_stream[inst_prob[i]], doc = inst_doc[i], site) -- what llda_sweep uses
 * with that stream_id and doc_base = 0 on the problem alone, so both paths leave identical states.
 * Arithmetic: one lane per ALLOWED topic (every instance of a launch has <= `lanes` allowed topics, lanes in
 * {8, 16, 32, 64}); the draw is decided from unnormalised fp64 prefix sums with a 2^-40 margin (it then equals the
 * reference pipeline's choice); a site that cannot be decided that way (~1e-11 per site) goes through
 * the reference's exact fp64 pipeline inside the kernel (status word 2 counts them).  Needs alpha, beta >= 1e-6,
 * V*beta < 2^40 (LLDA_E_BAD_ARG otherwise) and K_p <= 128 for every problem (the caller's responsibility). */
typedef struct llda_batch_args {
    const int64_t *inst_off;     /* [dev] [I+1] site offsets of all document instances               */
    const int32_t *order;        /* [dev] [n_inst] the instances this launch samples                  */
    int64_t        n_inst;
    const int32_t *word;         /* [dev] [S] word id per instance site                               */
    const int32_t *freq;         /* [dev] [S]                                                         */
    int32_t       *z;            /* [dev] [S] in/out device positions (layout of the instance's problem) */
    const int32_t *inst_prob;    /* [dev] [I] problem of every instance                               */
    const int32_t *inst_doc;     /* [dev] [I] index of the document inside its problem (RNG word 1)   */
    const int64_t *live_off;     /* [dev] [I+1] offsets into live_pos                                 */
    const int32_t *live_pos;     /* [dev] allowed device positions of every instance, in DRAW order
                                  * (ascending (lane, slot) of the instance's layout, see llda_sweep_args) */
    const int64_t *ndk_off;      /* [dev] [I] offset of the instance's n_dk row (KP_p entries) in n_dk */
    int32_t       *n_dk;         /* [dev] in/out                                                      */
    const int64_t *kw_off;       /* [dev] [P] offset of problem p's n_kw in counts / delta            */
    const int64_t *nk_off;       /* [dev] [P] offset of problem p's n_k                               */
    const int32_t *kp;           /* [dev] [P] KP_p                                                    */
    const int32_t *prob_stream;  /* [dev] [P] RNG counter word 2 of problem p                         */
    const int32_t *k;            /* [dev] [P] topics of problem p: 1 <= K_p <= 128 (one numpy pairwise leaf)  */
    const int32_t *counts;       /* [dev] sweep-start snapshot (read only)                            */
    int32_t       *delta;        /* [dev] += sweep changes                                            */
    int32_t       *status;       /* [dev] optional int32[4] as in llda_sweep_args                       */
    int64_t  V;
    int32_t  lanes;              /* lanes per instance: 8, 16, 32 or 64                               */
    int32_t  debug_margin;       /* 0 in production; n > 0 widens the margin to 2^-n; < 0: every site through the exact pipeline */
    double   alpha, beta;
    uint64_t seed;
    uint32_t sweep, reserved;
} llda_batch_args;
int llda_sweep_batch(const llda_batch_args *args, void *stream);

/* n_kw[i] += delta[i]; delta[i] = 0  for i < n   (end-of-sweep fold, after the all-reduce of delta). */
int llda_apply_delta(int32_t *counts, int32_t *delta, int64_t n, void *stream);

/* Fold a commit log into word-major counts (the deferred `+= f` / `-= f` of LabeledLDA.py:109-111,123-125).
 * The log is cut into ITEMS of consecutive entries of ONE word: item i covers log[item_begin[i] ..
 * item_begin[i] + item_len[i]) and belongs to word item_word[i] & 0x7fffffff; bit 31 of item_word is set when
 * the word is spread over several items (their rows are then combined with atomics, all others with plain
 * read-modify-writes).  One wavefront per item accumulates a KP-entry histogram in LDS and adds it to
 * target[word*KP ..].  target = n_kw (single device: counts updated in place) or n_kw_delta (zeroed; then
 * all-reduced and applied).  If n_k is not NULL the call also folds n_k += n_k_delta; n_k_delta = 0.
 * freq_csc[j] is the frequency of the site logged at j. */
int llda_commit_log(const int64_t *item_begin, const int32_t *item_len, const int32_t *item_word, int64_t n_items,
                    const uint32_t *commit_log, const int32_t *freq_csc, int32_t K, const int64_t *row_off,
                    int32_t *target, int32_t *n_k, int32_t *n_k_delta, void *stream);
/* row_off (dev, may be NULL = row v at target + v*KP): where the row of every word lives in `target`, for the
 * exchange between ranks.  row_off[v] >= 0: KP int32 at target + row_off[v].  row_off[v] < 0: KP/2 int32 at
 * target + ~row_off[v], each holding the counts of positions 2j (low half) and 2j+1 (high half) as int16 PAIRS;
 * counts are added as f * 65536^(p & 1), so words of several ranks can be summed with a plain int32 all-reduce
 * and decoded afterwards -- exact when the frequency mass of the word over ALL ranks and sites is <= 32767 (the
 * caller decides per word from that static bound; hot words stay int32).  Halves the bytes of the all-reduce.
 *
 * llda_apply_rows: counts[r*KP ..] += row r (pairs decoded), row zeroed, for r < n_rows.  With n_rows = V + 1
 * and counts = the fused [n_kw | n_k] buffer, row V is the n_k delta the sweep kernels wrote. */
int llda_apply_rows(const int64_t *row_off, int32_t *rows, int64_t n_rows, int32_t K, int32_t *counts, void *stream);

/* The 16-bit image of n_kw for llda_sweep_args.n_kw16 (ABI 15; K with llda_rows16_ok).  row16[v] != 0 flags the words whose
 * rows are packed: those whose counts can never leave 0 .. 65535 -- e.g. the words whose total over all topics, which Gibbs
 * sampling conserves, is at most 65535 (the reference's counts, LabeledLDA.py:109-111,123-125, only move a site's frequency
 * between two topics of one word).  Call it once per sweep, after the counts of the previous sweep were folded in and before
 * the first llda_sweep; unflagged rows of n_kw16 are not written.  A flagged row with a count outside the range sets bit 2
 * of status word 0. */
int llda_pack_rows16(const int32_t *n_kw, const uint8_t *row16, int64_t V, int32_t K, uint16_t *n_kw16, int32_t *status,
                     void *stream);

/* llda_pack_image with the image's columns in an order of the caller's (ABI 20): img[v][c] = min(n_kw[v][col_src[c]], 255 | 65535)
 * for c < KP; col_src (dev, int32[KP], 16-byte aligned) is a permutation of the device positions.  For llda_sweep_args.n_kw_img + img_col.
 * PRECONDITION, not checked on the host (the table lives on the device): col_src is a permutation of 0 .. KP-1 and img_col its inverse.
 * An entry outside 0 .. KP-1 is clamped to KP-1 by the kernels (no out-of-bounds access), a table that is not a permutation gives an
 * image that is not a copy of n_kw: the sweep's results are then undefined (no status bit reports it). */
int llda_pack_image_cols(const int32_t *n_kw, int64_t V, int32_t K, int32_t bits, const int32_t *col_src, void *img, void *stream);

/* The 16-bit image of EVERY row of n_kw plus, per word, whether all counts of its row fit 16 bits in this sweep's n_kw
 * (row16[v] = 1; ABI 19) -- for llda_sweep_args.row16.  K with llda_quad_ok (LLDA_E_BAD_K otherwise); n_kw and n_kw16 16-byte aligned.
 * Call it once per sweep, after the counts of the previous sweep were folded in and before the first llda_sweep.  The image of an
 * unflagged row holds the low halves of its counts and is never read. */
int llda_pack_rows16_all(const int32_t *n_kw, int64_t V, int32_t K, uint16_t *n_kw16, uint8_t *row16, void *stream);

/* The saturating narrow image of n counts for llda_sweep_args.n_kw_img (ABI 18): img[i] = min(n_kw[i], 255) as uint8_t
 * (bits 8) or min(n_kw[i], 65535) as uint16_t (bits 16); a negative count saturates as well.  n = V*KP, a multiple of 4;
 * n_kw 16-byte aligned, img 4-byte (bits 8) / 8-byte (bits 16) aligned -- llda_sweep asks the same of llda_sweep_args.n_kw_img.  Call it once per sweep, after the counts of the previous sweep were folded in and
 * before the first llda_sweep. */
int llda_pack_image(const int32_t *n_kw, int64_t n, int32_t bits, void *img, void *stream);

/* Count initialisation from assignments: LabeledLDA.py:89-92.  n_dk, n_kw, n_k must be zeroed by the
 * caller; z holds device positions.  (The SubLDA phantom-column quirk, CascadeLDA.py:382-385, is a
 * host-side initialisation and is uploaded as data.) */
int llda_count_init(const int64_t *doc_off, const int32_t *word, const int32_t *freq,
                    const int32_t *z, int64_t D, int32_t K,
                    int32_t *n_dk, int32_t *n_kw, int32_t *n_k, void *stream);

/* log-likelihood read-out (LabeledLDA.py:256-265): out_doc[d] (dev, double[D]) = sum over the sites
 * of document d of  -log( sum_k phi[k][w] * theta[d][k] ),  phi/theta as get_phi/get_theta
 * (LabeledLDA.py:231-239); sites are NOT weighted by frequency (reference quirk).
 * perplexity = exp( sum_d out_doc[d] / S ). */
int llda_loglik(const int64_t *doc_off, const int32_t *word, const uint16_t *lab_mask,
                const int32_t *n_dk, const int32_t *n_kw, const int32_t *n_k,
                int64_t D, int64_t V, int32_t K, double alpha, double beta,
                double *out_doc, void *stream);

/* Thinning read-outs with their running means, on the device (LabeledLDA.py:131-153, 231-239;
 * CascadeLDA.py:394-395, 423-434).  Outputs are in the REFERENCE's layout (row-major (K, V) and (D, K)
 * doubles), not the device permutation, so they can be handed to the caller as they are.
 *
 * llda_readout_phi:   cur[k][v] = (n_kw[v][k] + beta) / den[k]
 *     den == NULL: den[k] = n_k[k] + V*beta           -> get_phi   (LabeledLDA.py:231-234)
 *     den != NULL: double[KP] in device order, beta=0 -> SubLDA.get_ph with den = row sums of n_k_v
 *                                                        (CascadeLDA.py:394-395; 0/0 gives NaN as numpy does)
 * llda_readout_theta: num = n_dk[d] + labs[d]*alpha; cur[d] = num / np.sum(num)   (LabeledLDA.py:236-239;
 *     the row sum in numpy's pairwise order, bit for bit)
 * Both: mode 0: out = cur; mode 1: out = keep*out + share*cur, each product and the sum rounded
 * separately, as `factor*self.ph_hat + (1/s * cur_ph)` (LabeledLDA.py:144-145) and
 * `m * self.ph + (1-m) * cur_ph` (CascadeLDA.py:432) round -- the caller passes the two coefficients.
 * flags (dev int32[1], OR-ed, may be NULL; phi only) report the guards of LabeledLDA.py:146-153 on `out`:
 *   bit 0 an entry < 0, bit 1 a NaN, bit 2 a word whose column is all zero. */
#define LLDA_READOUT_NEGATIVE 1
#define LLDA_READOUT_NAN      2
#define LLDA_READOUT_NO_LOAD  4
int llda_readout_phi(const int32_t *n_kw, const int32_t *n_k, const double *den, int64_t V, int32_t K,
                     double beta, int32_t mode, double keep, double share, double *out, int32_t *flags,
                     void *stream);
int llda_readout_theta(const int32_t *n_dk, const uint16_t *lab_mask, int64_t D, int32_t K, double alpha,
                       int32_t mode, double keep, double share, double *out, void *stream);

/* Test-time fold-in sampler for held-out documents: one lane group per document, the topic-word
 * loadings are fixed and only the document's n_dk moves.  Covers
 *   LabeledLDA.prep4test + run_test        (LabeledLDA.py:155-212)
 *   CascadeLDA.prep4test + cascade_test    (CascadeLDA.py:186-247)
 *   CascadeLDA.prep4test + run_test        (CascadeLDA.py:299-344)
 * Per document: initial assignments drawn from the rows init_rows[init_idx[site]] (probabilities
 * prepared by the host exactly as the reference's prep4test does; RNG sweep word 0xFFFFFFFF) after
 * `while prob.sum() > 1: prob /= c_init`; then `iters` sweeps of
 *     prob = (n_dk + alpha) * ph[:, v];  prob /= prob.sum();
 *     [beta_fallback: if prob.sum() == 0 (0/0 raises in the reference): prob = (n_dk+alpha)*(ph[:,v]+beta), renormalise]
 *     while prob.sum() > 1: prob /= c_loop;   draw
 * Every `thinning` sweeps n_dk / sum(n_dk) enters a running average (avg_mode 0:
 * (s-1)/s*avg + (1/s)*cur; 1: m*avg + (1-m)*cur, m = (s-1)/s), written to th (D, KP).  z (device
 * positions) and n_dk (D, KP) receive the final state (iters = 0: the prep4test start state). */
typedef struct llda_foldin_args {
    const int64_t *doc_off;     /* [dev] [D+1]                                                  */
    const int32_t *word;        /* [dev] [S]                                                    */
    const int32_t *init_idx;    /* [dev] [S] row of init_rows for every site                     */
    const int32_t *freq;        /* [dev] [S]                                                    */
    const double  *ph;          /* [dev] [V*KP] loadings, word-major, device order, 0 in padding */
    const double  *init_rows;   /* [dev] [R*KP] initial probabilities, device order             */
    const uint8_t *slot_valid;  /* [dev] [KP] 1 where a slot holds a topic                       */
    int32_t *z;                 /* [dev] [S] out                                                 */
    int32_t *n_dk;              /* [dev] [D*KP] out                                              */
    double  *th;                /* [dev] [D*KP] out                                              */
    int32_t *status;            /* [dev] optional: bit 0 = a site had no positive probability    */
    int64_t D;
    int64_t doc_base;           /* RNG counter word 1 of document 0                              */
    int32_t K, iters, thinning, beta_fallback, avg_mode;
    int32_t exact_only;         /* 0 in production.  Test hook: 1 = every site through the reference's pipeline (numpy-ordered
                                   sum, IEEE divisions, shrink loop, keyed inverse-CDF draw) instead of deciding the draw from
                                   unnormalised prefix sums with a 2^-40 margin where that is provably the same topic */
    double alpha, beta, c_init, c_loop;
    uint64_t seed;
    uint32_t stream_id, reserved2;
    const int64_t *doc_ids;     /* [dev] [D] optional: RNG counter word 1 of every document (its low 32 bits);
                                   NULL = doc_base + d.  Lets documents with unrelated ids share one launch. */
    int64_t n_sites;            /* doc_off[D] (ABI 11).  > 0: the initial assignments are drawn by a separate launch with
                                   one lane group per SITE (they are independent of one another, and the reference's
                                   `while prob.sum() > 1: prob /= c` can run tens of thousands of times for one site);
                                   n_dk must then be zeroed by the caller.  0: inside the per-document launch. */
    const int64_t  *ph_base;    /* [dev] [D] optional (ABI 11): element offset of document d's loadings inside `ph` ...  */
    const uint32_t *doc_stream; /* [dev] [D] optional: ... and its RNG stream id (instead of stream_id): documents that are
                                   sampled against DIFFERENT label subsets of the same size (the nodes of one level of
                                   CascadeLDA.test_down_tree) share one launch                                        */
} llda_foldin_args;

int llda_foldin(const llda_foldin_args *args, void *stream);

/* Device self test of the kernel's division shortcut: runs >= n random (a, b) pairs through
 * "q = a * RN(1/b) + two exact-residual corrections" and through the hardware IEEE division and adds
 * the number of differing results to *mismatches_dev (dev, uint64, zeroed by the caller).  Expected: 0. */
int llda_selftest_div(uint64_t seed, int64_t n, unsigned long long *mismatches_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LLDA_GIBBS_H */
