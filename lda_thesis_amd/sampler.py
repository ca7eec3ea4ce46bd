"""Device-resident state of the collapsed Gibbs sampler and the sweep driver.

State mirrors the reference's sufficient statistics (/root/reference/LabeledLDA.py:73-79,
CascadeLDA.py:358-372) but lives in HBM as PyTorch-ROCm tensors in the group layout of
``layout.py``:

    n_kw  (V, KP) int32   <->  reference n_k_v (K, V) int64      (word-major + permuted)
    n_dk  (D, KP) int32   <->  reference n_d_k (D, K) int64
    n_k   (KP,)   int32   <->  reference n_zk  (K,)   int64
    z     (S,)    int32   <->  reference z_dn  (list of D int arrays), stored as device positions
    CSR corpus: doc_off (D+1) int64, word (S) int32, freq (S) int32   <->  docs / freqs lists
    lab_mask (D, G) int16 bit masks (bit s of lane g)                 <->  labs (D, K) float 0/1

One sweep = ``llda_sweep`` (HIP, include/llda_gibbs.h) under per-document snapshot semantics,
then -- when the documents are sharded over several GPUs -- one RCCL all-reduce (SUM, int32) of the
n_kw / n_k deltas over xGMI, then ``llda_apply_delta``.  Integer sums are exact and order
independent, so the state after a sweep is bit-identical for any number of GPUs.

Any K up to 7 688 (``layout.MAX_K``): up to 8 of numpy's pairwise-sum leaves the tuned "narrow" kernels run; beyond
that ``layout.wide`` is set, G counts the virtual lanes of one wavefront, and ``llda_sweep`` picks the wide kernels
 -- nothing in this class differs except the ``max_doc_tokens`` hint it hands them.
"""
import os

import numpy as np
import torch

from . import _native
from .layout import group_layout


def shard_documents(doc_off, world_size):
    """Contiguous document ranges balanced by site count: returns world_size+1 document bounds."""
    doc_off = np.asarray(doc_off, dtype=np.int64)
    D = doc_off.shape[0] - 1
    total = int(doc_off[-1])
    bounds = [0]
    for r in range(1, world_size):
        target = total * r // world_size
        b = int(np.searchsorted(doc_off, target, side="left"))
        bounds.append(min(max(b, bounds[-1]), D))
    bounds.append(D)
    return bounds


def _dist_active(group):
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


class GibbsSampler(object):
    """One shard of documents on one device.

    Parameters
    ----------
    doc_off, word, freq : CSR corpus of the LOCAL documents (numpy or torch).
    z        : topic id (reference numbering) of every local site.
    K, V     : topics (labels incl. root) and vocabulary size.
    labs     : None (all topics allowed: the dense mask of LocalLDA.py:60-84), a (D, K) 0/1 array
               (reference ``labs``), or a CSR pair (lab_off, lab_idx) of allowed topic ids.
    counts   : None -> counts are built on the device from z (LabeledLDA.py:89-92) and, when
               sharded, all-reduced; or a dict(n_d_k=(D,K), n_k_v=(K,V), n_zk=(K,)) in reference
               layout for the LOCAL documents / GLOBAL matrices (keeps e.g. SubLDA's phantom columns).
    seed, stream_id, doc_base : RNG key / counter words (doc_base = global id of local doc 0).
    group    : torch.distributed process group (None = default group when initialised).
    sparse_labels : True (default) = documents that allow few topics run through the sparse kernel when it
               fits (<= 64 allowed topics, at most a quarter of K); False forces the dense kernel.
    sharded  : True (default) = the local documents are one shard of a corpus spread over the ranks
               of ``group``: deltas are all-reduced every sweep.  False = a self-contained problem
               (e.g. one CascadeLDA sub-problem per GPU): no collective at all.
    commit_log : True = the sweep kernels log (old, new) topic per site in word-major order and
               ``llda_commit_log`` folds the log into n_kw without global atomics; False = int32 atomics on the
               delta buffer from inside the sweep kernels.  Same counts either way.  None (default) = log from
               2^20 local sites up (below that the extra pass costs more than the atomics it saves).
    overlap_ranges : C > 1 = the local documents are cut into C contiguous ranges; the exchange rows of range i are
               all-reduced (asynchronously, on the collective's own stream) while range i+1 is still being sampled.
               Every range exchanges its own dense rows, so C ranges move C times the bytes -- see DESIGN.md section 7
               for when that pays.  Needs the commit-log exchange rows on every rank (else it is ignored).
    rows16   : the sweep reads the n_kw rows of the words whose corpus-wide count fits 16 bits from a 16-bit image of
               n_kw that is refreshed at the start of every sweep (``llda_pack_rows16``): half the bytes per row, for a
               few more instructions per site.  Same results.  The int32 kernel of these layouts is bound by the fabric
               (L2 line fills), the 16-bit one by instruction issue, and since round 3's trimming the latter is the faster
               wherever the rows do not all sit in the L2s: + 10 % on BASELINE configs[3] (n_kw 205 MB), + 5 ... 8 % on Zipf
               corpora with V = 300 000 / 1 000 000, + 50 % with uniform words over a 1 GB n_kw, + 13 ... 21 % at K = 1024;
               - 6 % at K = 512 with a 41 MB n_kw, all at three waves per SIMD.  Where every document holds fewer than 2^16 tokens
               the kernel packs n_dk with its sweep-start value and runs FOUR waves per SIMD: another + 7 ... 10 %, and ahead of
               the int32 kernel at every size measured.  None (default) = where the kernel has it (dense
               mask, commit log, K = 512 or 1024) and either every document is below 2^16 tokens or n_kw is at least
               ROWS16_MIN_BYTES (64 MiB); True = wherever the
               kernel has it; False = off.  The image costs V*KP*2 bytes (half of n_kw again) inside the allocation of
               the counts plus 4 bytes per site; with rows16=None a shard that has no room for it sweeps with int32 rows
               (with a warning) and the environment variable LLDA_ROWS16=on|off decides for callers that cannot pass the
               argument.
    quad     : K = 512 (and every K whose layout has 16 slots per lane in 8, 16 or 32 lanes that all hold topics: 97 .. 128, most of
               185 .. 256 and 361 .. 512 -- ``_native.quad_ok``) with the 16-bit rows and every document below 2^16 tokens: the kernel
               that walks FOUR documents per wavefront (16 lanes x 32 slots each; eight / sixteen documents for the 16- / 8-lane layouts;
               csrc/kernel_quad.hpp) on an image of EVERY row -- which rows fit 16 bits is decided per
               sweep by ``llda_pack_rows16_all`` from the counts themselves; a row that does not is read as int32.  Same results.
               None (default) = wherever it applies AND the rows that do not fit 16 bits are rare: at most QUAD_MAX_WIDE_SITES of the
               sites may read such a row (the quad kernel reads it without prefetch) -- looked at when the sampler is built and every
               QUAD_CHECK_EVERY sweeps from the library's own flags, without synchronising; beyond that the sampler goes over to the
               two-document kernel with its int32 rows for good.  True = always; False = the two-documents-per-wavefront kernel;
               LLDA_QUAD=on|off in the environment decides for callers that cannot pass the argument.
    image_order : the narrow image keeps its columns in an order of its own: topics that are allowed TOGETHER (label co-occurrence over
               the local documents, weighted by their sites) are packed into the same 128-byte lines by a greedy clustering, so a site's
               gathers touch fewer lines -- the sparse-label kernel is bound by the L2's line fills.  The counts, the draw order and the
               results are untouched (``llda_pack_image_cols`` / ``llda_sweep_args.img_col``).  None (default) = taken when it saves at
               least a tenth of the lines a site touches (``image_lines_per_site`` = (before, after)); True = always; False = never.
    build_lock : a context manager the heavy LOCAL sections of the construction run under (the sorts of the commit log, the count
               initialisation, the images) -- never a collective.  For several processes that share one device (bench.py --one-device,
               tests): concurrent constructions are time-sliced by the GPU's scheduler and take seconds to minutes instead of one second.
    image    : sparse label sets (the sparse-label kernel): the kernel gathers its counts from a SATURATING narrow image of n_kw
               (8 or 16 bits per count, refreshed by ``llda_pack_image`` at the start of every sweep) and re-reads an entry that
               shows 255 / 65535 from n_kw itself: a row spans a quarter / half as many cache lines, and the kernel is bound by
               the L2's line fills.  Same results.  None (default) = 8, 16 or no image by the size of the problem (n_kw of at
               least IMAGE_MIN_BYTES and IMAGE_MIN_SITES sites) and the share of gathers that would escape; 0 = off; 8 / 16 = that
               image wherever the sparse-label kernel runs.  Costs V*KP (8) or 2*V*KP (16) bytes; LLDA_IMAGE=0|8|16 in the
               environment decides for callers that cannot pass the argument.
    """

    def __init__(self, doc_off, word, freq, z, K, V, alpha, beta, labs=None, counts=None, seed=0,
                 stream_id=0, doc_base=0, device=None, group=None, sort_docs=True,
                 docs_per_group=0, sharded=True, sparse_labels=True, commit_log=None, exchange_always=False,
                 overlap_ranges=1, rows16=None, image=None, quad=None, build_lock=None, image_order=None):
        _native.lib()                                       # fail loudly when the extension is missing
        _native.require_device()                            # ... or when no GPU is visible: there is no CPU fallback
        import contextlib
        locked = build_lock if build_lock is not None else contextlib.nullcontext()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.K, self.V = int(K), int(V)
        self.alpha, self.beta = float(alpha), float(beta)
        self.seed, self.stream_id, self.doc_base = int(seed), int(stream_id), int(doc_base)
        self.group = group
        self.sharded = bool(sharded)
        self.docs_per_group = int(docs_per_group)
        self.sweeps_done = 0
        self._status_event = self._status_host = None     # post_status: the asynchronous copy of the status word in flight
        self._wide_event = self._wide_host = self._word_sites = None    # _quad_policy
        self._parts_cache = {}
        self.debug_margin = 0          # test hook of the two-tier draw (include/llda_gibbs.h)
        self.kernel_events = None      # set to [] to record a (start, end) HIP event pair per sweep kernel
        self.exchange_always = bool(exchange_always)   # take the exchange path even with a single rank (tests)
        self.overlap_ranges = max(1, int(overlap_ranges))
        self.comm_events = None        # set to [] to record (start, end) event pairs around the waits for the collectives
        self.layout = lay = group_layout(self.K)
        dev = self.device

        def as_dev(a, dtype):
            if isinstance(a, torch.Tensor):
                return a.to(device=dev, dtype=dtype).contiguous()
            return torch.from_numpy(np.ascontiguousarray(np.asarray(a))).to(device=dev, dtype=dtype)

        self.doc_off = as_dev(doc_off, torch.int64)
        self.word = as_dev(word, torch.int32)
        self.freq = as_dev(freq, torch.int32)
        self.D = int(self.doc_off.shape[0] - 1)
        self.S = int(self.word.shape[0])
        if self.S:
            # device counts are int32 and the kernels move a site's count with a 24-bit multiply-add
            fmin, fmax, ftot = int(self.freq.min()), int(self.freq.max()), int(self.freq.sum(dtype=torch.int64))
            if fmin < 0 or fmax >= self.MAX_FREQ:
                raise ValueError("word frequencies inside a document must be in [0, %d); got %d .. %d" %
                                 (self.MAX_FREQ, fmin, fmax))
            if ftot >= (1 << 31):
                raise ValueError("the local shard holds %d tokens; the int32 device counts hold fewer than 2^31" % ftot)
            wmin, wmax = int(self.word.min()), int(self.word.max())
            if wmin < 0 or wmax >= self.V:
                raise IndexError("word id %d is out of bounds for a vocabulary of %d" % (wmax if wmax >= self.V else wmin, self.V))
        # wide layouts: the most tokens any document holds (llda_sweep_args.max_doc_tokens: below 2^15 the kernel keeps
        # int16 count changes in LDS instead of copies of the counts)
        self.max_doc_tokens = 0
        if lay.wide and self.S:
            pre = torch.zeros((self.S + 1,), dtype=torch.int64, device=dev)
            torch.cumsum(self.freq, 0, out=pre[1:])
            self.max_doc_tokens = int(min((pre[self.doc_off[1:]] - pre[self.doc_off[:-1]]).max().item(), 2 ** 31 - 1))
        self._topic_pos = torch.from_numpy(lay.topic_pos.astype(np.int64)).to(dev)
        self._pos_topic = torch.from_numpy(lay.pos_topic.astype(np.int64)).to(dev)
        self.z = self._topic_pos[as_dev(z, torch.int64)].to(torch.int32)

        self.lab_mask = self._make_masks(labs)
        # every topic allowed in every document (no labs, or label sets that are all complete): the kernels skip the mask
        self.dense_mask = labs is None or (self.D > 0 and bool((self.lab_mask == self._make_masks(None)[:1]).all()))
        # sparse label sets (Labeled LDA proper): positions of the allowed topics per document, ascending;
        # the library then runs one lane per ALLOWED topic (llda_sweep_sparse_kernel)
        self.live_off = self.live_pos = self._heavy = None
        self.live_max = 0
        if sparse_labels and labs is not None and self.D > 0:
            self._make_live()
        self.csc_pos = self.commit_log = self.site_rec = None
        self._ranges = self._make_ranges()
        if commit_log is None:
            commit_log = self.S >= (1 << 20)
        if self.S >= (1 << 31):
            commit_log = False                      # log positions are int32
        if commit_log and self.S > 0:
            with locked:
                self._make_commit_log()
        self._sort_docs = bool(sort_docs)
        self._call_limit = min(self.MAX_CALL_SITES, self.MAX_CALL_SITES_REC) if self.site_rec is not None else self.MAX_CALL_SITES
        self._off_host = self.doc_off.cpu().numpy() if (self.S > self._call_limit or len(self._ranges) > 2) else None
        self._calls = self._make_calls(self.doc_off[1:] - self.doc_off[:-1])
        self.doc_order = self._calls[0][2]       # (order of the first -- normally the only -- call)

        KP = lay.KP
        self.n_dk = torch.zeros((self.D, KP), dtype=torch.int32, device=dev)
        # n_kw and n_k (and their delta buffers) are two views of ONE allocation each, so that the per-sweep
        # exchange is a single all-reduce and the fold a single launch
        self._counts = torch.zeros(((self.V + 1) * KP,), dtype=torch.int32, device=dev)
        # (FLAG_TAIL more words behind every buffer that is all-reduced: the status flags of all ranks travel with the deltas, _stage_flags)
        self._delta_full = torch.zeros(((self.V + 1) * KP + self.FLAG_TAIL,), dtype=torch.int32, device=dev)
        self._delta = self._delta_full[:(self.V + 1) * KP]
        self._gflags = self._flag_bits_t = None     # the flags every rank agrees on, once a sweep has exchanged them
        self.n_kw = self._counts[:self.V * KP].view(self.V, KP)
        self.n_k = self._counts[self.V * KP:]
        self.n_kw_delta = self._delta[:self.V * KP].view(self.V, KP)
        self.n_k_delta = self._delta[self.V * KP:]
        self.status = torch.zeros((4,), dtype=torch.int32, device=dev)   # [flags, tier-0 unsure, exact tier, -]
        if counts is None:
            with locked:
                _native.count_init(self.doc_off, self.word, self.freq, self.z, self.D, self.K,
                                   self.n_dk, self.n_kw, self.n_k)
            if self.sharded and _dist_active(self.group):
                import torch.distributed as dist
                dist.all_reduce(self._counts, group=self.group)
        else:
            self.n_dk[:, self._topic_pos] = as_dev(counts["n_d_k"], torch.int32)
            self.n_kw[:, self._topic_pos] = as_dev(np.asarray(counts["n_k_v"]).T, torch.int32)
            self.n_k[self._topic_pos] = as_dev(counts["n_zk"], torch.int32)
        # wide layouts, dense or general label masks: work space that lets the sweep keep fp32 factors only in LDS
        # (llda_sweep_args.scratch)
        self._scratch = None
        if lay.wide and (self.live_off is None or self._heavy is not None) and self.D > 0:
            # (sparse label sets: the HEAVY documents of the shard go to the general wide kernel in a launch of their own, _lane_parts)
            nbytes = _native.sweep_scratch_bytes(self.K, max(hi - lo for lo, hi, _ in self._calls))
            if nbytes:
                self._scratch = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        self.row_off = self.rows = self._rows_list = None
        if self.sharded and (_dist_active(self.group) or exchange_always):
            self._make_exchange_rows()
        self.row16 = self.n_kw16 = self.site_row = None
        self.quad = False
        for var, allowed in (("LLDA_ROWS16", ("on", "off")), ("LLDA_QUAD", ("on", "off")), ("LLDA_IMAGE", ("0", "8", "16"))):
            if os.environ.get(var) is not None and os.environ[var] not in allowed:
                raise ValueError("%s=%r: expected one of %s" % (var, os.environ[var], ", ".join(allowed)))
        if rows16 is None and os.environ.get("LLDA_ROWS16") in ("on", "off"):     # for callers behind the LabeledLDA front end
            rows16 = os.environ["LLDA_ROWS16"] == "on"
        if quad is None and os.environ.get("LLDA_QUAD") in ("on", "off"):
            quad = os.environ["LLDA_QUAD"] == "on"
        self._quad_wanted = quad
        if (rows16 is not False and self.S and self.dense_mask and self.commit_log is not None
                and (_native.rows16_ok(self.K) or (quad is not False and _native.quad_ok(self.K)))
                and self.alpha >= 1e-6 and self.beta >= 1e-6):
            with locked:
                self._make_rows16(auto=rows16 is None)
        self.n_kw_img = None
        if image is None and os.environ.get("LLDA_IMAGE") in ("0", "8", "16"):   # for callers behind the LabeledLDA front end
            image = int(os.environ["LLDA_IMAGE"])
        if image not in (None, 0, 8, 16):
            raise ValueError("image must be None (automatic), 0 (off), 8 or 16")
        self._img_src = self._img_col = None
        self.image_lines_per_site = None
        if (image != 0 and self.S and self.live_off is not None and self.alpha >= 1e-6 and self.beta >= 1e-6
                and self.V * self.beta < 2.0 ** 40):
            self._make_image(image)
            if self.n_kw_img is not None and image_order is not False:
                with locked:
                    self._make_image_order(force=image_order)

    IMAGE_MIN_BYTES = 32 << 20       # image=None: below this n_kw (the eight L2s hold it) or below IMAGE_MIN_SITES sites the per-sweep
    IMAGE_MIN_SITES = 1 << 20        # llda_pack_image pass costs more than the line fills it saves
    IMAGE_MAX_ESCAPES = 0.5          # image=None: the narrowest image whose sampled escape rate stays below this (measured: with 35 % of
                                     # the gathers escaping -- the sparse variant of configs[3] at 1 M documents -- the 8-bit image is still
                                     # 18 % faster than the 16-bit one: the escapes go to the hot words' rows, which the L2s hold)

    def _image_escape_rates(self, sample=1 << 18):
        """share of the gathers of a sweep (site x allowed topic of its document) whose count would saturate an 8-bit / a 16-bit
        image, over an evenly strided sample of the local sites and the CURRENT counts -> (rate8, rate16)"""
        dev = self.device
        step = max(1, self.S // sample)
        sites = torch.arange(0, self.S, step, device=dev)
        docs = torch.bucketize(sites, self.doc_off[1:], right=True)
        lo, n = self.live_off[docs], (self.live_off[docs + 1] - self.live_off[docs])
        j = torch.arange(self.live_max, device=dev)
        ok = j[None, :] < n[:, None]
        pos = self.live_pos[torch.where(ok, lo[:, None] + j[None, :], lo[:, None]).clamp_(max=max(self.live_pos.numel() - 1, 0))].to(torch.int64)
        x = self.n_kw[self.word[sites].to(torch.int64)[:, None], pos]
        total = max(int(ok.sum().item()), 1)
        return (int(((x >= 255) & ok).sum().item()) / total, int(((x >= 65535) & ok).sum().item()) / total)

    def _make_image(self, bits=None):
        """the saturating narrow image of n_kw for the sparse-label kernels (llda_sweep_args.n_kw_img): uint8 / int16 [V, KP],
        refreshed by llda_pack_image at the start of every sweep.  bits=None picks 8, 16 or no image from the size of the problem
        and the sampled escape rates; the choice only ever changes how fast a sweep runs."""
        n = self.V * self.layout.KP
        if self.live_max == 0:
            return                                              # (every document is heavy: no launch of the sparse-label kernel)
        try:
            if bits is None:
                if n * 4 < self.IMAGE_MIN_BYTES or self.S < self.IMAGE_MIN_SITES:
                    return
                r8, r16 = self._image_escape_rates()
                bits = 8 if r8 <= self.IMAGE_MAX_ESCAPES else 16 if r16 <= self.IMAGE_MAX_ESCAPES else 0
                if not bits:
                    return
            self.n_kw_img = torch.zeros((n,), dtype=torch.uint8 if bits == 8 else torch.int16, device=self.device)
        except torch.cuda.OutOfMemoryError:
            import warnings
            warnings.warn("GibbsSampler: no room for the narrow image of n_kw (%.1f GB at 8 bits); gathering from n_kw itself" % (n / 1e9))
            self.n_kw_img = None

    def _make_image_order(self, force=None):
        """column order of the narrow image: a greedy clustering of the label co-occurrence matrix into lines of 128 bytes.
        C[p, q] = sum over the local documents that allow both p and q of the document's sites (every site gathers all of its
        document's topics); a line is seeded with the heaviest position not placed yet and filled with the positions that share the
        most weight with what the line already holds.  _img_src[c] = position held by image column c, _img_col = its inverse."""
        dev, KP = self.device, self.layout.KP
        per_line = 128 // self.n_kw_img.element_size()
        if KP <= per_line:
            return                                              # one line per row anyway
        lens = (self.doc_off[1:] - self.doc_off[:-1]).to(torch.float32)
        cnt = self.live_off[1:] - self.live_off[:-1]
        C = torch.zeros((KP, KP), dtype=torch.float64, device=dev)
        step = max(1, (1 << 26) // KP)
        for d0 in range(0, self.D, step):
            d1 = min(self.D, d0 + step)
            l0, l1 = int(self.live_off[d0]), int(self.live_off[d1])
            if l1 == l0:
                continue
            rows = torch.repeat_interleave(torch.arange(d1 - d0, device=dev), cnt[d0:d1])
            M = torch.zeros((d1 - d0, KP), dtype=torch.float32, device=dev)
            M[rows, self.live_pos[l0:l1].to(torch.int64)] = 1.0
            C += ((M * lens[d0:d1, None]).t() @ M).to(torch.float64)
        C = C.cpu().numpy()
        w = C.diagonal().copy()
        free = w > 0
        order = []
        while free.any():
            seed = int(np.argmax(np.where(free, w, -1.0)))
            free[seed] = False
            line, aff = [seed], C[seed].copy()
            while len(line) < per_line and free.any():
                cand = np.where(free, aff, -1.0)
                j = int(np.argmax(cand))
                if cand[j] <= 0.0:                              # nothing left that is ever allowed together with this line:
                    j = int(np.argmax(np.where(free, w, -1.0)))  # fill it with the heaviest of the rest rather than leave a hole
                free[j] = False
                line.append(j)
                aff += C[j]
            order.extend(line)
        rest = np.setdiff1d(np.arange(KP), np.array(order, dtype=np.int64), assume_unique=False)
        src = np.concatenate([np.array(order, dtype=np.int64), rest])
        col = np.empty(KP, dtype=np.int64)
        col[src] = np.arange(KP)
        col_t = torch.from_numpy(col).to(dev)
        # lines a site touches (distinct lines among its document's topics, weighted by the document's sites), before and after: the
        # order is only taken when it saves at least a tenth of them -- for label sets without structure (or already co-located by the
        # caller's label order) the plain pack is cheaper and the gathers no better
        docs = torch.repeat_interleave(torch.arange(self.D, device=dev), cnt)
        pos = self.live_pos[:int(self.live_off[-1])].to(torch.int64)

        def lines_per_site(column):
            n_lines = (KP + per_line - 1) // per_line
            key = torch.unique(docs * n_lines + column // per_line)
            per_doc = torch.bincount(key // n_lines, minlength=self.D).to(torch.float32)
            return float((per_doc * lens).sum().item() / max(float(lens.sum().item()), 1.0))
        self.image_lines_per_site = (lines_per_site(pos), lines_per_site(col_t[pos]))
        if force is not True and self.image_lines_per_site[1] > 0.9 * self.image_lines_per_site[0]:
            return
        self._img_src = torch.from_numpy(src.astype(np.int32)).to(dev)
        self._img_col = col_t.to(torch.int32)

    QUAD_MAX_WIDE_SITES = 0.02       # quad=None: largest share of the sites that may read a row which does not fit the 16-bit image
    QUAD_CHECK_EVERY = 32            # ... looked at every so many sweeps (asynchronously)

    def _quad_policy(self):
        """quad=None: the share of the sites whose row the library flagged as wide in THIS sweep's image, computed on the device and
        copied to the host asynchronously every QUAD_CHECK_EVERY sweeps; the copy is looked at -- after waiting for its event, which
        has long fired -- exactly QUAD_CHECK_EVERY sweeps later, so WHICH sweep hands over is a function of the counts and not of
        how fast the host runs (every run of the same corpus takes the same kernels; ranks decide on their own shard's share).
        Counts that concentrate (a frequent word settling in a few topics) take the sampler over to the two-document kernel.
        After a hand-over at K = 128 / 256 the 16-bit image stays allocated inside [n_kw | n_k | n_kw16] (half of n_kw again,
        unused from then on) along with site_rec and max_doc_tokens: the counts are never moved under a running sampler."""
        if self.sweeps_done % self.QUAD_CHECK_EVERY:
            return
        ev = self._wide_event
        if ev is not None:
            ev.synchronize()
            self._wide_event = None
            if float(self._wide_host[0]) > self.QUAD_MAX_WIDE_SITES * self.S:
                self.quad = False
                self.row16 = None
                if _native.rows16_ok(self.K):
                    self._flag_rows16()                # the static flags, bit 31 of csc_pos and site_row of the two-document kernel
                else:
                    self.n_kw16 = None                 # (K = 128, 256: the int32 rows of the general kernel; the image stays allocated)
                return
        if self._wide_host is None:
            self._wide_host = torch.zeros((1,), dtype=torch.float32).pin_memory()
        share = ((1.0 - self.row16.to(torch.float32)) * self._word_sites).sum().reshape(1)
        self._wide_host.copy_(share, non_blocking=True)
        self._wide_event = torch.cuda.Event()
        self._wide_event.record()

    ROWS16_MIN_BYTES = 64 << 20      # rows16=None, documents of 2^16 tokens or more (three waves per SIMD): below this n_kw
                                     # the L2s serve the int32 rows and the shorter kernel wins
    MAX_FREQ = 1 << 23   # v_mad_i32_i24 moves a site's count (include/llda_gibbs.h: freq)
    PAIR_LIMIT = 32767   # largest frequency mass of a word (all ranks) whose row is exchanged as int16 pairs

    def _make_rows16(self, auto=False):
        """16-bit rows (llda_sweep_args.n_kw16): an entry of n_kw can never exceed the total of its word's row -- the
        sweep only moves a site's frequency between two topics of one word (LabeledLDA.py:109-111,123-125) and the
        exchange sums such moves --, so the rows whose total is at most 65535 (all but the few hundred most frequent
        words of a natural corpus) can be read from a 16-bit image.  The flag of a site's word rides in bit 31 of its
        commit-log position, where the kernel sees it one site before it needs the row.

        The image lives in the SAME allocation as the counts, [n_kw | n_k | n_kw16]: llda_sweep_args.site_row addresses a
        row in 16-byte units from n_kw, so the distance between the two arrays is a constant of (V, KP) and never a
        property of where the caching allocator put two tensors."""
        V, KP = self.V, self.layout.KP
        if KP % 8:
            raise _native.NativeError("llda_rows16_ok(%d) holds but KP = %d is not a multiple of 8" % (self.K, KP))
        # site_row is an int32: the last 16-bit row starts (V+1)*KP/4 + (V-1)*KP/8 units after n_kw
        if (V + 1) * (KP // 4) + V * (KP // 8) >= 1 << 31:
            if auto:
                return
            raise ValueError("rows16=True: n_kw of %d x %d is too large for the 32-bit row offsets of the 16-bit-row kernel"
                             % (V, KP))
        # (the row sums of n_dk bound every entry and never change: a site moves its count between two topics)
        tokens_max = int(min(int(self.n_dk.sum(dim=1, dtype=torch.int64).max().item()), 2 ** 31 - 1)) if self.D else 0
        four_waves = 0 < tokens_max < 65536
        if auto and not four_waves and V * KP * 4 < self.ROWS16_MIN_BYTES:
            return
        quad = bool(self._quad_wanted is not False and four_waves and _native.quad_ok(self.K) and V < (1 << 22))
        two_doc = _native.rows16_ok(self.K)       # the kernel with static flags (bit 31 of csc_pos, site_row): K = 512, 1024
        if quad and self._quad_wanted is None:
            # sites whose word has a count beyond 16 bits somewhere in its row: rare, or the two-document kernel's prefetched int32 rows
            wide = (self.n_kw.max(dim=1).values > 65535) | (self.n_kw.min(dim=1).values < 0)
            self._word_sites = torch.bincount(self.word.to(torch.int64), minlength=V).to(torch.float32)
            if float((wide.to(torch.float32) * self._word_sites).sum().item()) > self.QUAD_MAX_WIDE_SITES * self.S:
                quad = False
        if self._quad_wanted and not quad:
            raise ValueError("quad=True: needs a K with llda_quad_ok (16 slots per lane in 8, 16 or 32 lanes: K = 100, 128, 200, 256, 400, 512 ...), "
                             "documents of fewer than 65 536 tokens and a vocabulary below 2^22 words")
        if not quad and not (two_doc and bool(self._rows16_fits().any())):
            return
        n32 = (V + 1) * KP
        try:
            both = torch.zeros((n32 + V * KP // 2,), dtype=torch.int32, device=self.device)
        except torch.cuda.OutOfMemoryError:
            if not auto:
                raise
            import warnings
            warnings.warn("GibbsSampler: no room for the 16-bit image of n_kw (%.1f GB); sweeping with int32 rows"
                          % (V * KP * 2 / 1e9))
            return
        both[:n32] = self._counts
        self._counts = both[:n32]
        self.n_kw = self._counts[:V * KP].view(V, KP)
        self.n_k = self._counts[V * KP:]
        self._counts16 = both                                  # (keeps the one allocation alive under its own name)
        self.n_kw16 = both[n32:].view(torch.int16)
        # four documents per wavefront (llda_sweep_args.row16): K = 512, documents below 2^16 tokens; the flags are the library's,
        # rewritten every sweep
        self.quad = quad
        if self.quad:
            self.row16 = torch.zeros((V,), dtype=torch.uint8, device=self.device)
            self.max_doc_tokens = tokens_max
            return
        self._flag_rows16()
        # llda_sweep_args.max_doc_tokens: below 2^16 the 16-bit-row kernel packs n_dk with its sweep-start value and runs four
        # waves per SIMD
        if not self.max_doc_tokens:
            self.max_doc_tokens = tokens_max

    def _rows16_fits(self):
        return (self.n_kw.sum(dim=1, dtype=torch.int64) <= 65535) & (self.n_kw.min(dim=1).values >= 0)

    def _flag_rows16(self):
        """which rows the sweep reads from the 16-bit image, from the row totals of the CURRENT n_kw: row16 (per word),
        bit 31 of csc_pos (per site) and site_row (per site; static until the totals change by other means than sweeps)."""
        KP = self.layout.KP
        fits = self._rows16_fits()
        self.row16 = fits.to(torch.uint8).contiguous()
        w = self.word.to(torch.int64)
        flagged = fits[w]
        self.csc_pos &= 0x7fffffff
        self.csc_pos |= torch.where(flagged, -(1 << 31), 0).to(torch.int32)
        gap = self.n_kw16.data_ptr() - self.n_kw.data_ptr()           # = (V+1)*KP*4 bytes: one allocation
        if gap != (self.V + 1) * KP * 4 or gap % 16:
            raise _native.NativeError("n_kw16 is not in the allocation of n_kw")
        self.site_row = torch.where(flagged, gap // 16 + w * (KP // 8), w * (KP // 4)).to(torch.int32).contiguous()

    def _make_exchange_rows(self):
        """Exchange layout of the per-sweep count deltas when every rank folds a commit log: one row per word plus
        the n_k row.  |delta[v][k]| can never exceed the frequency mass M_v of word v over ALL ranks (a static
        property of the corpus), so rows with M_v <= 32767 travel as int16 pairs packed in int32 words -- a plain
        int32 SUM all-reduce stays exact -- and only the hot words keep int32 rows: about half the bytes."""
        import torch.distributed as dist
        dev, KP = self.device, self.layout.KP
        mass = torch.zeros((self.V,), dtype=torch.int64, device=dev)
        if self.S:
            mass.index_add_(0, self.word.to(torch.int64), self.freq.to(torch.int64))
        everyone = torch.tensor([1 if (self.commit_log is not None or self.S == 0) else 0], device=dev)
        if _dist_active(self.group):
            dist.all_reduce(mass, group=self.group)
            dist.all_reduce(everyone, op=dist.ReduceOp.MIN, group=self.group)
        if int(everyone.item()) == 0:
            return                                   # some rank commits with atomics: int32 delta buffer for all
        pairs = mass <= self.PAIR_LIMIT
        size = torch.where(pairs, KP // 2, KP)
        size = torch.cat([size, torch.tensor([KP], device=dev)])          # row V: the n_k delta, int32
        off = torch.cumsum(size, 0) - size
        self.row_off = torch.where(torch.cat([pairs, torch.tensor([False], device=dev)]), ~off, off).contiguous()
        total = int(off[-1].item()) + KP
        # one buffer per overlap range (normally one): the rows of range i travel while range i+1 is sampled
        self._rows_full_list = [torch.zeros((total + self.FLAG_TAIL,), dtype=torch.int32, device=dev) for _ in range(len(self._ranges) - 1)]
        self._rows_list = [f[:total] for f in self._rows_full_list]
        self.rows = self._rows_list[0]
        # the sweep kernels add the n_k changes straight into row V; the int32 delta buffer is not needed
        self._nk_delta_list = [r[total - KP:] for r in self._rows_list]
        self.n_k_delta = self._nk_delta_list[0]
        self.n_kw_delta = None
        self._delta = self.rows
        self._delta_full = self._rows_full_list[0]

    def add_word_topic_counts(self, words, topics, amounts):
        """n_kw[word, topic] += amount for parallel 1-D arrays (host or device); n_k and n_dk are left alone.
        (SubLDA's phantom columns, reference CascadeLDA.py:382-385, are such counts.)"""
        dev = self.device
        w = torch.as_tensor(np.ascontiguousarray(words), dtype=torch.int64, device=dev)
        if w.numel() == 0:
            return
        pos = self._topic_pos[torch.as_tensor(np.asarray(topics), dtype=torch.int64, device=dev)]
        self.n_kw.index_put_((w, pos), torch.as_tensor(np.asarray(amounts), dtype=torch.int32, device=dev),
                             accumulate=True)
        if self.n_kw16 is not None and not self.quad:
            self._flag_rows16()            # row totals changed: a row may no longer fit 16 bits (or fit now)

    # ------------------------------------------------------------------ masks
    def _make_masks(self, labs):
        lay, dev = self.layout, self.device
        if labs is None:
            row = lay.lane_masks(np.ones((1, self.K)))[0].astype(np.int64)
            m = torch.from_numpy(row).to(dev).expand(self.D, lay.G)
        elif isinstance(labs, tuple):
            lab_off, lab_idx = labs
            lab_off = torch.as_tensor(np.asarray(lab_off), dtype=torch.int64, device=dev)
            lab_idx = torch.as_tensor(np.asarray(lab_idx).astype(np.int64), dtype=torch.int64, device=dev)
            counts = lab_off[1:] - lab_off[:-1]
            rows = torch.repeat_interleave(torch.arange(self.D, device=dev), counts)
            lane = torch.from_numpy(lay.topic_lane.astype(np.int64)).to(dev)[lab_idx]
            slot = torch.from_numpy(lay.topic_slot.astype(np.int64)).to(dev)[lab_idx]
            m = torch.zeros((self.D, lay.G), dtype=torch.int64, device=dev)
            m.index_put_((rows, lane), torch.ones_like(slot) << slot, accumulate=True)
        else:
            labs = np.asarray(labs)
            if labs.shape != (self.D, self.K):
                raise ValueError("labs must be (D, K) = (%d, %d)" % (self.D, self.K))
            m = torch.from_numpy(lay.lane_masks(labs).astype(np.int64)).to(dev)
        # stored as the uint16 bit pattern in an int16 tensor
        m = torch.where(m >= 32768, m - 65536, m).to(torch.int16).contiguous()
        return m

    def _make_live(self):
        """per document the device positions of its allowed topics (draw order) for the sparse-label kernel.  A document that allows
        more than 64 topics or more than a quarter of K is HEAVY: it keeps an empty list and is swept by the dense kernel with its
        label mask (its own launch, _lane_parts); when more than half of the documents are heavy the whole shard takes the dense
        kernel, as if sparse_labels were off."""
        lay, dev = self.layout, self.device
        shifts = torch.arange(lay.T, device=dev, dtype=torch.int32)
        lm_pos = torch.from_numpy(lay.lm_pos.astype(np.int64)).to(dev)
        counts, pos, heavy = [], [], []
        step = max(1, (1 << 26) // lay.KP)                 # documents per chunk: the (chunk, KP) 0/1 matrix stays small
        for d0 in range(0, self.D, step):
            bits = (self.lab_mask[d0:d0 + step].to(torch.int32) & 0xFFFF)          # (chunk, G) lane masks
            allowed = ((bits.unsqueeze(-1) >> shifts) & 1).reshape(bits.shape[0], lay.KP)   # (lane, slot) order = draw order
            c = allowed.sum(dim=1)
            h = (c > 64) | (c * 4 > self.K)
            allowed = allowed * (~h).to(allowed.dtype)[:, None]
            _, lm = torch.nonzero(allowed, as_tuple=True)                      # row-major => draw order ascending
            counts.append(torch.where(h, torch.zeros_like(c), c))
            heavy.append(h)
            pos.append(lm_pos[lm])                                             # ... as memory positions
        heavy = torch.cat(heavy)
        n_heavy = int(heavy.sum().item())
        if n_heavy * 2 > self.D:
            return                                                             # dense kernel is the better fit
        counts = torch.cat(counts)
        self.live_off = torch.zeros((self.D + 1,), dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=self.live_off[1:])
        self.live_pos = torch.cat(pos).to(torch.int32).contiguous()
        if self.live_pos.numel() == 0:                                         # (a pointer the library can test)
            self.live_pos = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.live_max = int(counts.max().item())
        self._heavy = heavy if n_heavy else None

    def _make_ranges(self):
        """document bounds of the overlap ranges (contiguous, balanced by site count); one range = no overlap."""
        C = self.overlap_ranges
        if C <= 1:
            return [0, self.D]
        return shard_documents(self.doc_off.cpu().numpy(), C)      # (empty ranges when D < C: every rank makes C)

    LOG_ITEM = 4096    # most log entries one wavefront of llda_commit_log folds (hot words are cut into items)
    MAX_CALL_SITES = (1 << 30) - 1   # llda_sweep addresses the sites of one call with 32-bit byte offsets
    MAX_CALL_SITES_REC = (1 << 28) - 1   # ... and the 16-byte site records of narrow layouts

    def _make_calls(self, lens):
        """document ranges of the llda_sweep calls of one sweep (one range unless the shard spans 2^30 sites),
        each with its processing order: longest documents first, neighbours in a wavefront of similar length."""
        def order(lo, hi):
            ln = lens[lo:hi]
            if not self._sort_docs or hi - lo < 2 or int(ln.min()) == int(ln.max()):
                return None
            return torch.sort(ln, descending=True, stable=True).indices.to(torch.int32)
        if self.S <= self._call_limit and len(self._ranges) == 2:
            return [(0, self.D, order(0, self.D))]
        off = self._off_host
        if self.D and int(np.diff(off).max()) > self._call_limit:
            raise ValueError("a document has more than %d sites" % self._call_limit)
        calls = []
        for r in range(len(self._ranges) - 1):
            lo, end = self._ranges[r], self._ranges[r + 1]
            while lo < end:
                hi = int(np.searchsorted(off, off[lo] + self._call_limit, side="right")) - 1
                hi = max(min(hi, end), lo + 1)
                calls.append((lo, hi, order(lo, hi)))
                lo = hi
        return calls or [(0, 0, None)]                       # (a rank without documents)

    def _lane_parts(self, lo, hi, order):
        """the launches of one llda_sweep call: [(doc_order, documents, live_max)]; live_max 0 = the dense kernel with the label masks.
        Dense masks, or sparse label sets of at most 8 topics per document: one launch.  Otherwise the sparse-label kernel gives every
        document as many lanes as the LARGEST label set of its launch needs (8, 16, 32 or 64: 8 ... 1 documents per wavefront), so the
        documents are split by the lanes THEY need -- a corpus in which a few documents carry twenty labels and the rest a handful no
        longer runs all of them one to a half-wavefront -- and the HEAVY documents (_make_live) go to the dense kernel.  Which launch a
        document is in changes nothing (snapshot semantics); ``order`` is kept inside a class."""
        if self.live_off is None or (self.live_max <= 8 and self._heavy is None) or hi <= lo:
            return [(order, hi - lo, self.live_max)]
        key = (lo, hi)
        hit = self._parts_cache.get(key)
        if hit is not None and hit[0] is order:              # (the SAME tensor object: an address can be handed out again)
            return hit[1]
        idx = order.to(torch.int64) if order is not None else torch.arange(hi - lo, device=self.device)
        n = (self.live_off[lo + 1:hi + 1] - self.live_off[lo:hi])[idx]
        heavy = self._heavy[lo:hi][idx] if self._heavy is not None else torch.zeros_like(n, dtype=torch.bool)
        parts = []
        for lanes, low in ((8, -1), (16, 8), (32, 16), (64, 32)):             # (a document that allows nothing: first class, as before)
            sel = idx[(n > low) & (n <= lanes) & ~heavy]
            if sel.numel():
                parts.append((sel.to(torch.int32).contiguous(), int(sel.numel()), lanes))
        sel = idx[heavy]
        if sel.numel():
            parts.append((sel.to(torch.int32).contiguous(), int(sel.numel()), 0))
        self._parts_cache[key] = (order, parts)              # (holds the keyed tensor: one entry per call range)
        return parts

    def _make_commit_log(self):
        """word-major (CSC) view of the sites -- range by range when the exchange is pipelined over document ranges
        -- : position of every site, frequencies in that order, and the work items of llda_commit_log (runs of at
        most LOG_ITEM entries of one word of one range)."""
        dev, V = self.device, self.V
        C = len(self._ranges) - 1
        w64 = self.word.to(torch.int64)
        if C > 1:                                            # key = range * V + word
            site_bounds = self.doc_off[torch.as_tensor(self._ranges[1:-1], dtype=torch.int64, device=dev)]
            rng = torch.bucketize(torch.arange(self.S, device=dev), site_bounds, right=True)
            w64 = w64 + rng * V
        order = torch.sort(w64, stable=True).indices
        self.csc_pos = torch.empty((self.S,), dtype=torch.int32, device=dev)
        self.csc_pos[order] = torch.arange(self.S, dtype=torch.int32, device=dev)
        self.freq_csc = self.freq[order].contiguous()
        del order
        per_word = torch.bincount(w64, minlength=C * V)
        word_off = torch.zeros((C * V + 1,), dtype=torch.int64, device=dev)
        torch.cumsum(per_word, 0, out=word_off[1:])
        n_items = (per_word + (self.LOG_ITEM - 1)) // self.LOG_ITEM            # 0 for words without a site
        words = torch.repeat_interleave(torch.arange(C * V, device=dev), n_items)
        first = torch.zeros((C * V + 1,), dtype=torch.int64, device=dev)
        torch.cumsum(n_items, 0, out=first[1:])
        part = torch.arange(words.numel(), device=dev) - first[words]          # index of the item within its word
        self.item_begin = (word_off[words] + part * self.LOG_ITEM).contiguous()
        self.item_len = torch.minimum(per_word[words] - part * self.LOG_ITEM,
                                      torch.full_like(part, self.LOG_ITEM)).to(torch.int32).contiguous()
        shared = n_items[words] > 1
        wid = words % V
        self.item_word = torch.where(shared, wid - (1 << 31), wid).to(torch.int32).contiguous()
        self._item_bounds = first[::V].cpu().tolist()                          # items of range r: [b[r], b[r+1])
        self.commit_log = torch.zeros((self.S,), dtype=torch.int32, device=dev)
        # layouts with 8 or 16 lanes per document: {word, freq, csc_pos} as one 16-byte record per site
        # (llda_sweep_args.site_rec: those kernels are bound by the number of cache lines their scalar loads touch)
        self.site_rec = None
        if self.layout.G <= 16:
            self.site_rec = torch.stack([self.word, self.freq, self.csc_pos, torch.zeros_like(self.word)], dim=1).contiguous()

    # ------------------------------------------------------------------ the hot path
    def _timed(self, fn):
        """run fn(); with comm_events enabled on a GPU, bracket it with events on the current stream (the time the
        compute stream spends in / waiting for the collective = the exposed communication)."""
        if self.comm_events is None or self.device.type != "cuda":
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self.comm_events.append((a, b))
        return out

    def sweep(self):
        """One Gibbs sweep over the local documents + exchange + fold."""
        import torch.distributed as dist
        ev = None
        if self.kernel_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()              # same stream the kernel is enqueued on (torch's current stream)
        logged = self.commit_log is not None
        have_group = dist.is_available() and dist.is_initialized()
        exchange = self.sharded and (_dist_active(self.group) or self.exchange_always)
        n_ranges = len(self._ranges) - 1
        if self.quad:                                         # this sweep's n_kw: nothing below changes it before the fold
            _native.pack_rows16_all(self.n_kw, self.K, self.n_kw16, self.row16)
            if self._quad_wanted is None:
                self._quad_policy()
        if self.quad:
            pass
        elif self.n_kw16 is not None:
            _native.pack_rows16(self.n_kw, self.row16, self.K, self.n_kw16, self.status)
        if self.n_kw_img is not None and self._img_src is not None:
            _native.pack_image_cols(self.n_kw, self.K, self._img_src, self.n_kw_img)
        elif self.n_kw_img is not None:
            _native.pack_image(self.n_kw, self.n_kw_img)
        # decided from values every rank agrees on (a rank with an empty shard logs nothing but must still issue
        # one collective per range)
        pipelined = exchange and self.rows is not None and n_ranges > 1
        if len(self._calls) == 1:
            self._calls[0] = (0, self.D, self.doc_order)
        works, call = [], 0
        for r in range(n_ranges):
            nk_delta = self._nk_delta_list[r] if pipelined else self.n_k_delta
            while call < len(self._calls) and self._calls[call][0] < self._ranges[r + 1]:
                lo, hi, order = self._calls[call]
                call += 1
                s0 = 0 if len(self._calls) == 1 else int(self._off_host[lo])
                s1 = self.S if len(self._calls) == 1 else int(self._off_host[hi])
                # (sparse label sets: one launch per class of lanes a document needs -- see _lane_parts)
                for part, n_docs, live_max in self._lane_parts(lo, hi, order):
                    sparse = self.live_off is not None and live_max > 0
                    _native.sweep(doc_off=self.doc_off[lo:hi + 1], doc_order=part, word=self.word, freq=self.freq,
                                  z=self.z, lab_mask=self.lab_mask[lo:hi], n_dk=self.n_dk[lo:hi], n_kw=self.n_kw,
                                  n_kw_delta=self.n_kw_delta, n_k=self.n_k, n_k_delta=nk_delta,
                                  status=self.status, D=n_docs, V=self.V, K=self.K, alpha=self.alpha,
                                  beta=self.beta, seed=self.seed, sweep=self.sweeps_done,
                                  stream_id=self.stream_id, doc_base=self.doc_base + lo,
                                  docs_per_group=self.docs_per_group, dense_mask=self.dense_mask,
                                  debug_margin=self.debug_margin,
                                  live_off=self.live_off[lo:hi + 1] if sparse else None,
                                  live_pos=self.live_pos if sparse else None,
                                  live_max=live_max, csc_pos=self.csc_pos, commit_log=self.commit_log,
                                  n_sites=s1 - s0, site_rec=self.site_rec, max_doc_tokens=self.max_doc_tokens,
                                  scratch=self._scratch, n_kw16=self.n_kw16, site_row=self.site_row,
                                  n_kw_img=self.n_kw_img if sparse else None, row16=self.row16 if self.quad else None,
                                  img_col=self._img_col if sparse and self.n_kw_img is not None else None)
            if pipelined:
                # fold this range's log into ITS exchange rows and start their all-reduce: it runs on the
                # collective's stream (ordered after the fold) while the next range is sampled on this one
                i0, i1 = (self._item_bounds[r], self._item_bounds[r + 1]) if logged else (0, 0)
                if i1 > i0:
                    _native.commit_log(self.item_begin[i0:i1], self.item_len[i0:i1], self.item_word[i0:i1],
                                       self.commit_log, self.freq_csc, self.K, self._rows_list[r],
                                       row_off=self.row_off)
                if have_group:
                    last = r == n_ranges - 1                  # (the flags of every range's kernels are there once the last one is queued)
                    if last:
                        self._stage_flags(self._rows_full_list[r])
                    works.append(dist.all_reduce(self._rows_full_list[r] if last else self._rows_list[r], group=self.group, async_op=True))
        if ev is not None:
            ev[1].record()
            self.kernel_events.append(ev)
        if not exchange:
            if logged:
                # single device: the log is folded straight into n_kw (and n_k += its delta) -- no delta pass
                _native.commit_log(self.item_begin, self.item_len, self.item_word, self.commit_log,
                                   self.freq_csc, self.K, self.n_kw, self.n_k, self.n_k_delta)
            else:
                _native.apply_delta(self._counts, self._delta)
        elif pipelined:
            self._timed(lambda: [w.wait() for w in works])
            if have_group:
                self._take_flags(self._rows_full_list[-1])
            for r in range(n_ranges):
                _native.apply_rows(self.row_off, self._rows_list[r], self.K, self._counts)
        elif self.rows is not None:
            # every rank folds its log into the exchange rows (int16 pairs for all but the hot words), ONE int32
            # all-reduce over xGMI, then the rows are decoded into [n_kw | n_k]
            if logged:
                _native.commit_log(self.item_begin, self.item_len, self.item_word, self.commit_log,
                                   self.freq_csc, self.K, self.rows, row_off=self.row_off)
            if have_group:
                self._stage_flags(self._rows_full_list[0])
                self._timed(lambda: dist.all_reduce(self._rows_full_list[0], group=self.group))
                self._take_flags(self._rows_full_list[0])
            _native.apply_rows(self.row_off, self.rows, self.K, self._counts)
        else:
            if logged:
                _native.commit_log(self.item_begin, self.item_len, self.item_word, self.commit_log,
                                   self.freq_csc, self.K, self.n_kw_delta)
            if have_group:
                self._stage_flags(self._delta_full)
                self._timed(lambda: dist.all_reduce(self._delta_full, group=self.group))   # RCCL over xGMI: SUM int32, one collective
                self._take_flags(self._delta_full)
            _native.apply_delta(self._counts, self._delta)
        self.sweeps_done += 1

    def comm_stats(self):
        """mean exposed communication per sweep (ms the compute stream spent in / waiting for the collectives) from
        the recorded comm_events, with the exchange geometry; None when nothing was recorded."""
        if not self.comm_events:
            return None
        ms = [a.elapsed_time(b) for a, b in self.comm_events]
        n_coll = len(self._rows_list) if self._rows_list is not None else 1
        nbytes = (self.rows.numel() if self.rows is not None else self._delta.numel()) * 4
        return {"exposed_ms_per_sweep": float(np.mean(ms)), "collectives_per_sweep": n_coll,
                "bytes_per_collective": int(nbytes), "overlap_ranges": len(self._ranges) - 1}

    def exchange_description(self):
        """what travels between the GPUs per sweep (for reports)."""
        if self.rows is not None:
            pairs = int((self.row_off[:-1] < 0).sum().item())
            return ("one RCCL int32 SUM all-reduce per sweep of the n_kw / n_k deltas as exchange rows: %d of %d word rows "
                    "as int16 pairs (frequency mass <= %d), %.1f MB" % (pairs, self.V, self.PAIR_LIMIT,
                                                                          self.rows.numel() * 4 / 1e6))
        return "one RCCL int32 SUM all-reduce per sweep of the n_kw / n_k delta buffer, %.1f MB" % (self._delta.numel() * 4 / 1e6)

    STATUS_EVERY = 4     # sweeps between two asynchronous copies of the status word (post_status)
    FLAG_TAIL = 2        # words behind an exchanged buffer: how many ranks have status bit 0 / bit 2 set
    _FLAG_BITS = (1, 4)

    # The status word is written by the kernels of ONE rank; a rank that raised on its own flags while the others went on into the
    # next collective would leave them waiting for the process-group timeout.  So the two flags that raise travel with the deltas
    # (SUM of 0 / 1 per rank in the tail of the exchanged buffer: no collective of their own), and check_status / post_status look at
    # the SUMMED flags: every rank sees the same word after the same sweep and raises, or does not, together.
    def _stage_flags(self, full):
        if self._flag_bits_t is None:
            self._flag_bits_t = torch.tensor(self._FLAG_BITS, dtype=torch.int32, device=full.device)
        full[-self.FLAG_TAIL:] = ((self.status[:1] & self._flag_bits_t) != 0).to(torch.int32)

    def _take_flags(self, full):
        if self._gflags is None:
            self._gflags = torch.zeros((self.FLAG_TAIL,), dtype=torch.int32, device=full.device)
        self._gflags.copy_(full[-self.FLAG_TAIL:])
        full[-self.FLAG_TAIL:].zero_()

    def _status_word(self):
        """one-element device tensor: the status flags -- after an exchange, those of ALL ranks (identical everywhere)."""
        if self._gflags is None:
            return self.status[:1]
        g = (self._gflags > 0).to(torch.int32)
        return (g[:1] * 1 + g[1:2] * 4) | (self.status[:1] & 2)

    def post_status(self, every=None):
        """For callers that loop ``sweep()`` themselves (the reference's ``training_iteration`` API, LabeledLDA.py:101): every
        ``every`` sweeps an asynchronous copy of the status word is queued behind the sweep, and the copy queued EARLIER is looked at
        when it has landed -- no synchronisation; a site without a topic of positive probability (numpy raises at that site,
        LabeledLDA.py:117-119) raises here at most ``every`` + 1 sweeps late instead of only at the next thinning point."""
        import torch
        if self.device.type != "cuda":                   # (host-logic tests drive this class with CPU tensors and a stand-in backend)
            return
        ev = self._status_event
        if ev is not None and ev.query():
            self._status_event = None
            self._raise_for(int(self._status_host[0]))
        every = self.STATUS_EVERY if every is None else every
        if self._status_event is None and self.sweeps_done % max(1, every) == 0:
            if self._status_host is None:
                self._status_host = torch.zeros((4,), dtype=torch.int32).pin_memory()
            self._status_host[:1].copy_(self._status_word(), non_blocking=True)
            self._status_event = torch.cuda.Event()
            self._status_event.record()

    def check_status(self):
        """Raise like the reference would (numpy's multinomial rejects a NaN pvals vector)."""
        self._raise_for(int(self._status_word().item()))

    @staticmethod
    def _raise_for(st):
        if st & 1:
            raise ValueError("a site had no topic with positive probability (pvals would be NaN)")
        if st & 4:
            raise RuntimeError("a count left 0 .. 65535 where it is kept in 16 bits (a flagged row of n_kw, or an entry of n_dk "
                               "under the four-wave kernel): the counts handed to the sampler do not belong to its corpus")

    # ------------------------------------------------------------------ read-outs
    def loglik_sum(self):
        """sum over local sites of -log(phi[:, w] . theta_d)  (LabeledLDA.py:256-265), on device."""
        out = torch.zeros((self.D,), dtype=torch.float64, device=self.device)
        _native.loglik(self.doc_off, self.word, self.lab_mask, self.n_dk, self.n_kw, self.n_k,
                       self.D, self.V, self.K, self.alpha, self.beta, out)
        return float(out.sum().item())

    def perplexity(self):
        total, sites = self.loglik_sum(), float(self.S)
        if self.sharded and _dist_active(self.group):
            import torch.distributed as dist
            t = torch.tensor([total, sites], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, group=self.group)
            total, sites = float(t[0]), float(t[1])
        return float(np.exp(total / sites))

    # ------------------------------------------------------------------ thinning read-outs on the device
    # (K, V) / (D, K) float64 tensors in the REFERENCE's layout.  With ``out`` and the two coefficients the
    # call updates a running mean in place: out = keep*out + share*current.
    def phi(self, out=None, keep=None, share=None, flags=None):
        """get_phi: (n_k_v + beta) / (n_zk[:, None] + V*beta), reference LabeledLDA.py:231-234."""
        if out is None:
            out = torch.empty((self.K, self.V), dtype=torch.float64, device=self.device)
        _native.readout_phi(self.n_kw, self.n_k, None, self.V, self.K, self.beta, out, flags, keep, share)
        return out

    def ph_rows(self, out=None, keep=None, share=None):
        """SubLDA.get_ph: n_k_v / n_k_v.sum(axis=1) without smoothing (reference CascadeLDA.py:394-395); the
        row sums are exact integers."""
        if out is None:
            out = torch.empty((self.K, self.V), dtype=torch.float64, device=self.device)
        den = self.n_kw.sum(dim=0, dtype=torch.int64).to(torch.float64)
        _native.readout_phi(self.n_kw, None, den, self.V, self.K, 0.0, out, None, keep, share)
        return out

    def theta(self, out=None, keep=None, share=None):
        """get_theta of the LOCAL documents: (n_d_k + labs*alpha) / row sums, reference LabeledLDA.py:236-239."""
        if out is None:
            out = torch.empty((self.D, self.K), dtype=torch.float64, device=self.device)
        _native.readout_theta(self.n_dk, self.lab_mask, self.D, self.K, self.alpha, out, keep, share)
        return out

    # ------------------------------------------------------------------ reference-layout views
    def n_d_k(self):
        return self.n_dk[:, self._topic_pos].cpu().numpy().astype(np.int64)

    def n_k_v(self):
        return self.n_kw[:, self._topic_pos].t().contiguous().cpu().numpy().astype(np.int64)

    def n_zk(self):
        return self.n_k[self._topic_pos].cpu().numpy().astype(np.int64)

    def z_topics(self):
        """flat array of topic ids (reference numbering) of the local sites."""
        return self._pos_topic[self.z.to(torch.int64)].cpu().numpy().astype(np.int64)

    def z_dn(self):
        z = self.z_topics()
        off = self.doc_off.cpu().numpy()
        return [z[off[d]:off[d + 1]].copy() for d in range(self.D)]


class HostOrDevice(object):
    """A float64 matrix that is either on the host (numpy, what the reference's attributes are) or on the
    device (while the thinning read-outs accumulate into it); never both, so neither copy can go stale."""

    def __init__(self, host):
        self.host, self.dev = host, None

    def get(self, gather=None):
        if self.host is None:
            local = self.dev.cpu().numpy()
            self.host, self.dev = (gather(local) if gather is not None else local), None
        return self.host

    def set(self, value):
        self.host, self.dev = value, None

    def on_device(self, device, shape, fresh=False, rows=None):
        """the device copy (uploading the host one, or its ``rows`` slice, unless ``fresh``)."""
        if self.dev is None:
            if fresh:
                self.dev = torch.empty(shape, dtype=torch.float64, device=device)
            else:
                h = self.host if rows is None else self.host[rows[0]:rows[1]]
                self.dev = torch.from_numpy(np.ascontiguousarray(h, dtype=np.float64)).to(device)
            self.host = None
        return self.dev
