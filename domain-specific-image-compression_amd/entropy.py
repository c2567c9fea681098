"""Mirror of code/modelv2/eval_selfcontained_entropy.py: custom_compress /
custom_decompress / evaluate-style helpers, running on the GPU.

The reference script is a sketch that cannot execute (StudentT.cdf is not
implemented in torch; torchac is called with arguments it does not accept —
SURVEY.md §8c).  This module keeps its interface — the returned dict of :68-74,
`tail=10`, z string then y string per image, decode order of :76-123 — and the
interpretation frozen in DESIGN.md "Entropy path".
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import lib as _lib
from . import ops
from .ops import _f32c, _p, _stream

DEFAULT_LMAX = 256


class EntropyError(RuntimeError):
    pass


def _check_err(err, what):
    code = int(err.item())
    if code:
        reasons = [m for bit, m in ((1, "support wider than Lmax"), (2, "symbol outside its support"),
                                    (4, "output capacity exceeded")) if code & bit]
        raise EntropyError(f"{what}: " + ", ".join(reasons))


def gaussian_cdf(x):
    """:14-15 in float32, like the reference's CPU torch evaluation (host, through the
    library's own table math: x / float(sqrt 2), erf, 1 + ., 0.5 * .)."""
    L = _lib.load()
    a = np.asarray(x, dtype=np.float32)
    return np.array([L.dsic_host_gaussian_cdf_f32(float(v)) for v in a.ravel()], dtype=np.float32).reshape(a.shape)


def pmf_to_uint16_cdf(pmf):
    """:17-23 on the host: pmf [L, C, ...] float32 (support axis first) -> uint16 [L+1, C, ...]."""
    L = _lib.load()
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    Ls = p.shape[0]
    C = int(np.prod(p.shape[1:], dtype=np.int64)) if p.ndim > 1 else 1
    out = np.empty((Ls + 1, C), dtype=np.uint16)
    _lib.check(L.dsic_host_pmf_to_uint16_cdf(p.ctypes.data_as(ctypes.c_void_p), Ls, C,
                                             out.ctypes.data_as(ctypes.c_void_p)), "pmf_to_uint16_cdf")
    return out.reshape((Ls + 1,) + p.shape[1:])


def sigma_z_of(model):
    """:32 `torch.exp(model.z_prior.log_sigma)` (no clamp), evaluated on the host by the
    library's deterministic exp (float32 of the float64 value) so that an encoder and a decoder
    on different machines build the same z tables; returned on the model's device."""
    L = _lib.load()
    ls = model.z_prior.log_sigma.detach()
    vals = np.array([L.dsic_host_exp_f32(float(v)) for v in ls.cpu().numpy().astype(np.float32).ravel()],
                    dtype=np.float32)
    return torch.from_numpy(vals).to(ls.device)


def latent_support(y_tilde, z_tilde, tail=10):
    """-> meta int32 [B,4] = (ymin-tail, Ly, zmin-tail, Lz) on the device (:39-41, :52-54)."""
    y = _f32c(y_tilde, "latent_support")
    z = _f32c(z_tilde, "latent_support")
    B = y.shape[0]
    meta = torch.empty((B, 4), dtype=torch.int32, device=y.device)
    _lib.check(_lib.load().dsic_latent_support(_p(y), _p(z), _p(meta), B, y[0].numel(), z[0].numel(), int(tail),
                                               _stream()), "latent_support")
    return meta


def cdf_tables(sigma_y, nu_y, sigma_z, meta, Lmax=DEFAULT_LMAX, err=None):
    """-> (tab_y [B,rows,Lmax], tab_z [B,N,Lmax]) uint16 coder tables (:43-47, :55-61, :17-23).

    sigma_y/nu_y: [B,M] (one row per channel) or [B,M,Hy,Wy] (spatial_params: one row per latent
    element, NCHW order = symbol order)."""
    sy = _f32c(sigma_y, "cdf_tables").reshape(sigma_y.shape[0], -1)
    ny = _f32c(nu_y, "cdf_tables").reshape(nu_y.shape[0], -1)
    sz = _f32c(sigma_z, "cdf_tables")
    B, M = sy.shape
    N = sz.numel()
    dev = sy.device
    if err is None:
        err = torch.zeros(1, dtype=torch.int32, device=dev)
    # entries k >= L_b of a row are never read (every reader stops at the support width in `meta`)
    tab_y = torch.empty((B, M, Lmax), dtype=torch.uint16, device=dev)
    tab_z = torch.empty((B, N, Lmax), dtype=torch.uint16, device=dev)
    L = _lib.load()
    _lib.check(L.dsic_cdf_tables_gauss(_p(sz), _p(meta), _p(tab_z), B, N, Lmax, _p(err), _stream()),
               "cdf_tables_gauss")
    _lib.check(L.dsic_cdf_tables_student(_p(sy), _p(ny), _p(meta), _p(tab_y), B, M, Lmax, _p(err), _stream()),
               "cdf_tables_student")
    return tab_y, tab_z, err


def _cap(n):
    return (2 * n + 16 + 3) // 4 * 4   # <= 16 bits per symbol + flush, multiple of 4


@torch.no_grad()
def compress_latents(y_tilde, z_tilde, sigma_y, nu_y, sigma_z, tail=10, Lmax=DEFAULT_LMAX, streams_per_wg=1):
    """Device-resident compress of already computed latents.

    y_tilde [B,M,Hy,Wy], z_tilde [B,N,Hz,Wz] integer-valued (quant_mode="round");
    sigma_y/nu_y [B,M]; sigma_z [N].  Returns dict with device tensors:
    bytes uint8 [B, cap_z+cap_y], lengths int32 [B,2] (z,y), meta int32 [B,4],
    cap_z, cap_y, tab_y, tab_z, err.  Nothing synchronises with the host.
    """
    y = _f32c(y_tilde, "compress")
    z = _f32c(z_tilde, "compress")
    B, M, Hy, Wy = y.shape
    _, N, Hz, Wz = z.shape
    dev = y.device
    per_element = sigma_y.dim() == 4        # spatial_params: a table row per latent element
    meta = latent_support(y, z, tail)
    tab_y, tab_z, err = cdf_tables(sigma_y, nu_y, sigma_z, meta, Lmax)
    cap_y, cap_z = _cap(M * Hy * Wy), _cap(N * Hz * Wz)
    # the coder ORs its bits in: zero-filled, as 32-bit words (a byte fill kernel takes 4x the elements)
    out = torch.zeros((B, (cap_z + cap_y) // 4), dtype=torch.int32, device=dev).view(torch.uint8)
    lengths = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().dsic_range_encode(_p(y), _p(z), _p(meta), _p(tab_y), _p(tab_z), Lmax, B, M, Hy * Wy,
                                             N, Hz * Wz, _p(out), cap_y, cap_z, _p(lengths), _p(err),
                                             int(streams_per_wg), int(per_element), _stream()), "range_encode")
    return {"bytes": out, "lengths": lengths, "meta": meta, "cap_z": cap_z, "cap_y": cap_y,
            "tab_y": tab_y, "tab_z": tab_z, "err": err, "shape_y": list(y.shape), "shape_z": list(z.shape)}


def masked_streams(coder_cus=16, total_cus=256, xcds=8):
    """(main, coder) torch ExternalStreams on disjoint CU sets.

    The coder gets coder_cus/xcds CUs of EVERY XCD (workgroups are dealt round-robin over the
    XCDs, so removing CUs from one XCD only would make it the straggler of every conv launch).
    The pattern k*32+j with j = k, k+8, ... selects the same number of CUs per XCD whether mask
    bits enumerate CUs XCD-major or XCD-interleaved."""
    L = _lib.load()
    words = (total_cus + 31) // 32
    per_xcd = max(1, coder_cus // xcds)
    bits = np.zeros(total_cus, dtype=bool)
    for k in range(xcds):
        for t in range(per_xcd):
            bits[k * (total_cus // xcds) + (k + 8 * t) % (total_cus // xcds)] = True

    def make(sel):
        m = np.zeros(words, dtype=np.uint32)
        for i in np.nonzero(sel)[0]:
            m[i // 32] |= np.uint32(1 << (i % 32))
        h = ctypes.c_void_p()
        _lib.check(L.dsic_stream_create_masked(m.ctypes.data_as(ctypes.c_void_p), words, ctypes.byref(h)),
                   "stream_create_masked")
        return torch.cuda.ExternalStream(h.value)

    return make(~bits), make(bits)


class AsyncCompressor:
    """Runs compress_latents on side HIP streams so that the serial range coder overlaps
    synthesis (and the following batches' analysis).  Use as the `after_rate` hook of
    CompressionModel.forward; call .wait() before reading `.last`.

    `depth` side streams are used round-robin: the coder of batch i+1 does not queue behind the
    coder of batch i, so a coder that takes longer than one step (512x512 patches: 196 608 y symbols
    per string, a serial chain) still keeps up - its latency is hidden, its throughput doubles."""

    def __init__(self, model, tail=10, Lmax=DEFAULT_LMAX, stream=None, streams_per_wg=1, depth=1):
        self.model, self.tail, self.Lmax = model, tail, Lmax
        self.streams_per_wg = streams_per_wg
        self.streams = [stream if stream is not None else torch.cuda.Stream()]
        self.streams += [torch.cuda.Stream() for _ in range(max(1, int(depth)) - 1)]
        self.stream = self.streams[0]
        self.calls = 0
        self.last = None
        self._sigma_z = None
        self.timing = False          # bench.py: keep a HIP-event pair per call on the coder's stream
        self.times = []
        self._events = []            # pre-created timing events (see reserve_events)
        self._ready = None

    def reserve_events(self, n):
        """create n (start, done) timing-event pairs now, so that none is created while timing"""
        with torch.cuda.stream(self.stream):
            while len(self._events) < n:
                pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                pair[0].record(self.stream)
                pair[1].record(self.stream)
                self._events.append(pair)

    def _pair(self):
        if self._events:
            return self._events.pop()
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def __call__(self, partial):
        main = torch.cuda.current_stream()
        side = self.streams[self.calls % len(self.streams)]
        self.calls += 1
        if self._sigma_z is None:
            self._sigma_z = sigma_z_of(self.model)
            for st in self.streams:
                self._sigma_z.record_stream(st)
        if self._ready is None:
            self._ready = torch.cuda.Event()
        ready = self._ready              # re-recorded per call; wait_event captures this record
        ready.record(main)
        tensors = [partial["y_tilde"], partial["z_tilde"], partial["sigma"], partial["nu"]]
        with torch.cuda.stream(side):
            side.wait_event(ready)
            for t in tensors:
                t.record_stream(side)             # allocator must not recycle them under the coder
            if self.timing:
                e0, done = self._pair()
                e0.record(side)
            else:
                done = torch.cuda.Event()
            self.last = compress_latents(tensors[0], tensors[1], tensors[2], tensors[3], self._sigma_z,
                                         self.tail, self.Lmax, self.streams_per_wg)
            done.record(side)
            if self.timing:
                self.times.append((e0, done))
            self.last["done"] = done

    def wait(self):
        for st in self.streams:
            torch.cuda.current_stream().wait_stream(st)
        return self.last


def _per_channel(t):
    """[B,M,H,W] spatially constant (expanded) -> contiguous [B,M]; per-element tensors
    (spatial_params) and [B,M] pass through."""
    if t.dim() == 4 and t.stride(2) == 0 and t.stride(3) == 0:
        return t[:, :, 0, 0].contiguous()
    return t.contiguous()


def _tight_lmax(meta, minimum=32):
    """Support width of this batch (host sync): keeps per-element tables small."""
    L = int(meta[:, [1, 3]].max().item())
    return max(minimum, (L + 7) // 8 * 8)


TABLE_FLOW_VERSION = 2     # 1: float64 tables (round 1); 2: float32 CPU-torch operation order


def numerics_tag() -> int:
    """What a decoder must share with the encoder to rebuild the same coder tables: the table flow, and the
    arithmetic of the kernels that recompute sigma / nu from the decoded z (h_s: code/modelv2/
    eval_selfcontained_entropy.py:100-106).  bits 0-7 table flow version, bit 8 split-bf16 Winograd kernels
    (DSIC_WINO_BF16), bit 9 Winograd at all (DSIC_WINOGRAD), bits 16-31 the library's ABI version.  A stream carries
    the tag of its encoder; custom_decompress refuses a stream whose tag is not its own."""
    from . import layers as _layers
    return (TABLE_FLOW_VERSION | (int(bool(_layers.WINO_BF16)) << 8) | (int(bool(_layers.USE_WINOGRAD)) << 9)
            | ((int(_lib.load().dsic_abi_version()) & 0xFFFF) << 16))


@torch.no_grad()
def custom_compress(model, x, tail=10, Lmax=DEFAULT_LMAX):
    """eval_selfcontained_entropy.py:26-74.  Returns the reference's dict:
    strings [[z_bytes, y_bytes], ...], shape_y, shape_z, min_y, max_y, min_z, max_z."""
    out = model(x, quant_mode="round")
    sigma_z = sigma_z_of(model)                                        # :32 (no clamp)
    if getattr(model, "spatial_params", False):
        # one table row per latent element: size the rows to the actual support (the reference
        # reads min/max on the host here too, :39-40,52-53)
        Lmax = min(1000, _tight_lmax(latent_support(out["y_tilde"], out["z_tilde"], tail)))
    while True:
        c = compress_latents(out["y_tilde"], out["z_tilde"], _per_channel(out["sigma"]),
                             _per_channel(out["nu"]), sigma_z, tail, Lmax)
        try:
            _check_err(c["err"], "custom_compress")
            break
        except EntropyError:
            if Lmax >= 1000 or not (int(c["err"].item()) & 1):
                raise
            Lmax = min(1000, Lmax * 2)                                 # wider support than expected
    lengths = c["lengths"].cpu().numpy()
    meta = c["meta"].cpu().numpy()
    raw = c["bytes"].cpu().numpy()
    strings = []
    for b in range(raw.shape[0]):
        zs = raw[b, :lengths[b, 0]].tobytes()
        ys = raw[b, c["cap_z"]:c["cap_z"] + lengths[b, 1]].tobytes()
        strings.append([zs, ys])
    return {
        "strings": strings,
        "shape_y": c["shape_y"], "shape_z": c["shape_z"],
        "min_y": [int(m[0]) for m in meta], "max_y": [int(m[0] + m[1] - 1) for m in meta],
        "min_z": [int(m[2]) for m in meta], "max_z": [int(m[2] + m[3] - 1) for m in meta],
        "numerics": numerics_tag(),        # beyond the reference's keys (:68-74): see numerics_tag()
    }


def _upload_strings(strings, which, dev):
    lens = [len(s[which]) for s in strings]
    stride = max(4, (max(lens) + 3) // 4 * 4)
    buf = np.zeros((len(strings), stride), dtype=np.uint8)
    for b, s in enumerate(strings):
        buf[b, :lens[b]] = np.frombuffer(s[which], dtype=np.uint8)
    return (torch.from_numpy(buf).to(dev), torch.tensor(lens, dtype=torch.int32, device=dev), stride)


@torch.no_grad()
def custom_decompress(model, compressed, Lmax=None):
    """eval_selfcontained_entropy.py:76-123: decode z, re-run h_s, decode y, run g_s, clamp."""
    dev = next(model.parameters()).device
    tag = compressed.get("numerics")
    if tag is not None and int(tag) != numerics_tag():
        raise EntropyError(f"custom_decompress: the stream was written with numerics tag {int(tag):#x}, this decoder "
                           f"is {numerics_tag():#x} (table flow / kernel arithmetic differ: the coder tables would "
                           "not match and the latents would decode to garbage)")
    strings = compressed["strings"]
    B = len(strings)
    _, M, Hy, Wy = compressed["shape_y"]
    _, N, Hz, Wz = compressed["shape_z"]
    meta_np = np.array([[compressed["min_y"][b], compressed["max_y"][b] - compressed["min_y"][b] + 1,
                         compressed["min_z"][b], compressed["max_z"][b] - compressed["min_z"][b] + 1]
                        for b in range(B)], dtype=np.int32)
    if Lmax is None:
        Lmax = int(meta_np[:, [1, 3]].max())
        Lmax = (Lmax + 7) // 8 * 8 if getattr(model, "spatial_params", False) else max(DEFAULT_LMAX, Lmax)
    meta = torch.from_numpy(meta_np).to(dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    L = _lib.load()
    sigma_z = sigma_z_of(model)
    tab_z = torch.zeros((B, N, Lmax), dtype=torch.uint16, device=dev)
    _lib.check(L.dsic_cdf_tables_gauss(_p(sigma_z), _p(meta), _p(tab_z), B, N, Lmax, _p(err), _stream()),
               "cdf_tables_gauss")
    zbuf, zlen, zstride = _upload_strings(strings, 0, dev)
    z_hat = torch.empty((B, N, Hz, Wz), dtype=torch.float32, device=dev)
    _lib.check(L.dsic_range_decode(_p(zbuf), zstride, _p(zlen), 1, 0, _p(meta), 2, _p(tab_z), Lmax, B, N,
                                   Hz * Wz, 0, _p(z_hat), _p(err), _stream()), "range_decode(z)")
    # :100-106: hyper-synthesis on the decoded z
    (_, _, sigma_y, nu_y), _ = model.h_s.params_nhwc(ops.nchw_to_nhwc(z_hat), model.min_nu, model.max_nu)
    per_element = int(getattr(model, "spatial_params", False))
    rows = M * Hy * Wy if per_element else M
    tab_y = torch.zeros((B, rows, Lmax), dtype=torch.uint16, device=dev)
    _lib.check(L.dsic_cdf_tables_student(_p(sigma_y.contiguous()), _p(nu_y.contiguous()), _p(meta), _p(tab_y), B,
                                         rows, Lmax, _p(err), _stream()), "cdf_tables_student")
    ybuf, ylen, ystride = _upload_strings(strings, 1, dev)
    y_hat = torch.empty((B, M, Hy, Wy), dtype=torch.float32, device=dev)
    _lib.check(L.dsic_range_decode(_p(ybuf), ystride, _p(ylen), 1, 0, _p(meta), 0, _p(tab_y), Lmax, B, M,
                                   Hy * Wy, per_element, _p(y_hat), _p(err), _stream()), "range_decode(y)")
    _check_err(err, "custom_decompress")
    x_hat = model.g_s.forward_nhwc(ops.nchw_to_nhwc(y_hat))            # :120
    return x_hat.clamp(0, 1)                                           # :123


def real_bpp(compressed, H, W):
    """:148-149 — 8 * total bytes / (H*W), per batch."""
    total_bits = sum(len(s) * 8 for entry in compressed["strings"] for s in entry)
    return total_bits / float(H * W)


# ---- container: the reference keeps the compressed patch batch as an in-memory dict (:68-74);
# this is that dict as one byte string, so the strings can leave the process ------------------
_MAGIC = b"DSIC2\x00"      # DSIC1: rounds 1-2, no numerics tag (float64 / float32 tables both wrote it: not decodable
_MAGIC_V1 = b"DSIC1\x00"   # safely any more, refused)


def pack_container(compressed) -> bytes:
    """dict of custom_compress -> bytes.  Layout (little endian):
    magic(6) | numerics tag (uint32, numerics_tag()) | B,My,Hy,Wy,Nz,Hz,Wz (7 x uint32) | per image:
    min_y,max_y,min_z,max_z (4 x int32), len_z,len_y (2 x uint32) | per image: z bytes, y bytes."""
    import struct
    B, My, Hy, Wy = compressed["shape_y"]
    _, Nz, Hz, Wz = compressed["shape_z"]
    tag = int(compressed.get("numerics", numerics_tag()))
    head = [_MAGIC, struct.pack("<I", tag), struct.pack("<7I", B, My, Hy, Wy, Nz, Hz, Wz)]
    body = []
    for b in range(B):
        zs, ys = compressed["strings"][b]
        head.append(struct.pack("<4i2I", compressed["min_y"][b], compressed["max_y"][b], compressed["min_z"][b],
                                compressed["max_z"][b], len(zs), len(ys)))
        body += [zs, ys]
    return b"".join(head + body)


def unpack_container(blob: bytes):
    """Inverse of pack_container."""
    import struct
    if blob[:6] == _MAGIC_V1:
        raise ValueError("DSIC1 container: written before the numerics tag existed (its coder tables may be the float64 "
                         "ones of round 1); re-encode")
    if blob[:6] != _MAGIC:
        raise ValueError("not a DSIC container")
    (tag,) = struct.unpack_from("<I", blob, 6)
    B, My, Hy, Wy, Nz, Hz, Wz = struct.unpack_from("<7I", blob, 10)
    off = 10 + 28
    meta = [struct.unpack_from("<4i2I", blob, off + 24 * b) for b in range(B)]
    off += 24 * B
    strings = []
    for m in meta:
        zs = blob[off:off + m[4]]
        off += m[4]
        ys = blob[off:off + m[5]]
        off += m[5]
        strings.append([zs, ys])
    if off != len(blob):
        raise ValueError("truncated or oversized DSIC container")
    return {"strings": strings, "shape_y": [B, My, Hy, Wy], "shape_z": [B, Nz, Hz, Wz],
            "min_y": [m[0] for m in meta], "max_y": [m[1] for m in meta],
            "min_z": [m[2] for m in meta], "max_z": [m[3] for m in meta], "numerics": tag}
