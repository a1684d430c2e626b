"""NumPy's legacy normal stream, generated on the device: `np.random.normal(0, s, shape)` of the global RandomState
(`apply_gaussian_noise`, /root/reference/transformation.py:273-275; `TransformationPool.gaussian_noise`,
pipenline/cifar_image_transformations.py:39-48) for the SAME seed gives the SAME numbers — without the host drawing
them (10 ms per 375 x 500 image, which was most of the drivers' wall time).

What NumPy does (numpy/random/src/legacy/legacy-distributions.c, mt19937.c; restated):
  * MT19937: 624 state words; word k of the output stream is the TEMPERED word k of the state sequence; a block of 624
    new state words is computed from the previous block (`imgxf_mt19937_blocks`, one workgroup, sequential in the block
    index) — the raw state sequence is kept, so the generator's state at any position of the stream is simply a slice;
  * legacy_double: a = next >> 5, b = next >> 6, (a * 2^26 + b) / 2^53;
  * legacy_gauss (polar Box-Muller): x1 = 2 u - 1, x2 = 2 u' - 1, r2 = x1^2 + x2^2, REJECTED unless 0 < r2 < 1; else
    f = sqrt(-2 log(r2) / r2); the call returns f x2 and keeps f x1 for the next call (has_gauss);
  * normal = loc + scale * gauss, then the caller's .astype(float32).
Groups of four words are independent: all of them are evaluated at once, a prefix sum of the acceptance flags says
which ones the sequential loop would have used, and the position of the last one used is where the stream — and with it
the generator state handed back to NumPy — continues.

Exactness: every step is integer arithmetic or a correctly rounded IEEE operation except `log`, where the device's and
glibc's may differ in the last bit.  The float32 result can only differ if the double lies within 2^-46 (relative) of a
float32 rounding boundary — one sample in two million; those are recomputed with the host's libm from their exact (x, r2)
and patched in, as is the cached second normal that an odd-length draw hands back to NumPy as generator state.
"""
from __future__ import annotations

import math
import os
from typing import List, Sequence

import numpy as np
import torch

ACCEPT = 0.7853981633974483            # pi / 4: the probability that a group of four words is accepted
_TWO53 = 9007199254740992.0


def temper(y: torch.Tensor) -> torch.Tensor:
    """MT19937 tempering of raw state words (int64 tensor holding uint32 values)."""
    y = y ^ (y >> 11)
    y = y ^ ((y << 7) & 0x9D2C5680)
    y = y ^ ((y << 15) & 0xEFC60000)
    y = y ^ (y >> 18)
    return y & 0xFFFFFFFF


def mt_next_block(key: np.ndarray) -> np.ndarray:
    """mt19937_gen restated in NumPy (the test oracle of imgxf_mt19937_blocks): the 624 state words after `key`."""
    old = key.astype(np.uint64)
    new = np.zeros(624, np.uint64)

    def twist(u, v):
        y = (u & 0x80000000) | (v & 0x7FFFFFFF)
        return (y >> 1) ^ np.where(v & 1, 0x9908B0DF, 0).astype(np.uint64)
    new[:227] = old[397:624] ^ twist(old[:227], old[1:228])
    new[227:454] = new[0:227] ^ twist(old[227:454], old[228:455])
    new[454:623] = new[227:396] ^ twist(old[454:623], old[455:624])
    new[623] = new[396] ^ twist(old[623:624], new[0:1])[0]
    return (new & 0xFFFFFFFF).astype(np.uint32)


def words_needed(n_normals: int) -> int:
    """Stream words that certainly hold `n_normals` normals: the expected number of four-word groups plus a margin of
    more than 12 standard deviations (a multiple of 4)."""
    groups = (n_normals + 1) // 2
    want = int(groups / ACCEPT * 1.01) + 3000
    return 4 * want


class Draw:
    """The result of `normals`: float32 noise tensors (on the stream tensor's device), the stream position after the last
    draw, the generator's pending cached value, and whether any sample was too close to a float32 rounding boundary."""
    __slots__ = ("noise", "position", "has_gauss", "gauss", "patched")


def _host_gauss(x: float, r2: float) -> float:
    """f x with f = sqrt(-2 log(r2) / r2) in the host's libm, as legacy_gauss computes it."""
    return math.sqrt(-2.0 * math.log(r2) / r2) * x


MARGIN = 2.0 ** -46          # relative; the device's log is within an ulp (2^-53) of glibc's, f x and the scaling add a few more


def normals(raw: torch.Tensor, start: int, has_gauss: bool, gauss: float, requests: Sequence[tuple], f64: bool = False) -> Draw:
    """`raw`: the raw MT19937 state sequence as an int32 / int64 tensor of uint32 bit patterns (block 0 = the generator's
    current key; what imgxf_mt19937_blocks writes), `start`: the generator's position in it (its `pos`), `has_gauss` /
    `gauss`: its cached normal.  `requests`: (count, scale) per draw, in the order NumPy would be called.  Raises ValueError
    if `raw` is too short for the margin of words_needed.

    legacy_gauss is a STREAM of normals — the pairs of the accepted groups in order, a cached value being the second of a
    pair — and consecutive np.random.normal calls just take consecutive stretches of it: all requests are served by ONE pass
    over the words (no per-request synchronisation), then cut at the known counts and scaled.

    Exactness of the float32 results: a sample whose double lies within MARGIN (relative) of a float32 rounding boundary —
    one in 2^21, a few per million — is recomputed with the HOST's log from its exact (x, r2) and patched in; so is the
    cached normal that an odd number of normals leaves behind (it becomes generator state).
    `f64=True` returns the DOUBLES (TransformationPool.gaussian_noise adds them to the float32 image in double and truncates,
    cifar_image_transformations.py:39-48): a pixel's byte can only depend on the last bits of the double when the noise is
    within 1e-9 of an integer; those samples take the host path instead."""
    dev = raw.device
    pos = int(start)
    odt = torch.float64 if f64 else torch.float32
    counts = [int(n) for n, _ in requests]
    total = sum(counts)
    d = Draw()
    if total == 0:
        d.noise = [torch.empty((0,), dtype=odt, device=dev) for _ in requests]
        d.position, d.has_gauss, d.gauss, d.patched = pos, has_gauss, float(gauss), 0
        return d
    n2 = total - (1 if has_gauss else 0)                     # normals to take from new groups
    if raw.is_cuda and not f64 and raw.dtype == torch.int32 and os.environ.get("IMGXF_NP_FUSED", "1") != "0":
        return _normals_fused(raw, pos, has_gauss, float(gauss), requests, counts, total, n2)
    xs = ar = None
    if n2 > 0:
        groups = (n2 + 1) // 2
        w = words_needed(n2)
        if pos + w > raw.numel():
            raise ValueError("the MT19937 stream is shorter than the draw's margin")
        t = temper(raw[pos:pos + w].to(torch.int64) & 0xFFFFFFFF).view(-1, 4)
        u1 = ((t[:, 0] >> 5).double() * 67108864.0 + (t[:, 1] >> 6).double()) / _TWO53
        u2 = ((t[:, 2] >> 5).double() * 67108864.0 + (t[:, 3] >> 6).double()) / _TWO53
        del t
        x1, x2 = 2.0 * u1 - 1.0, 2.0 * u2 - 1.0
        del u1, u2
        r2 = x1 * x1 + x2 * x2
        acc = (r2 < 1.0) & (r2 != 0.0)
        rank = torch.cumsum(acc, 0)
        last = int((rank < groups).sum().item())             # index of the groups-th accepted group
        if last >= rank.numel():
            raise ValueError("too few accepted groups inside the margin")                      # (> 12 sigma: not expected to happen)
        del rank
        sel = acc[:last + 1]
        a1, a2, ar = x1[:last + 1][sel], x2[:last + 1][sel], r2[:last + 1][sel]
        del x1, x2, r2, acc, sel
        f = torch.sqrt(-2.0 * torch.log(ar) / ar)
        xs = torch.stack((a2, a1), 1).reshape(-1)            # the call returns f x2 first, f x1 on the next call
        del a1, a2
        vals = (f.repeat_interleave(2) * xs)[:n2]
        del f
        pos += 4 * (last + 1)
    # the stream the requests cut up: [cached normal (exact, host libm)] + vals; sample e of vals is (xs[e], ar[e // 2])
    lead = 1 if has_gauss else 0
    out: List[torch.Tensor] = []
    patched, at = 0, 0
    for count, scale in requests:
        count, scale = int(count), float(scale)
        if count == 0:
            out.append(torch.empty((0,), dtype=odt, device=dev))
            continue
        lo, hi = at - lead, at + count - lead                # range in vals (lo = -1: the cached normal comes first)
        head = []
        if lo < 0:
            v = 0.0 + scale * float(gauss)
            head, lo = [v if f64 else np.float32(v)], 0
        if hi > lo:
            nd = 0.0 + scale * vals[lo:hi]                   # legacy_normal: loc + scale * gauss
            if f64:
                res = nd
                risky = ((nd - torch.round(nd)).abs() < 1e-9).nonzero().flatten()
            else:
                res = nd.float()
                risky = ((nd * (1.0 - MARGIN)).float() != (nd * (1.0 + MARGIN)).float()).nonzero().flatten()
            if risky.numel():
                e = risky + lo
                xv, rv = xs[e].cpu().tolist(), ar[e // 2].cpu().tolist()
                exact = [0.0 + scale * _host_gauss(x, r) for x, r in zip(xv, rv)]
                res[risky] = torch.tensor(exact if f64 else [np.float32(v) for v in exact], dtype=odt).to(dev)
                patched += int(risky.numel())
            out.append(torch.cat((torch.tensor(head, dtype=odt, device=dev), res)) if head else res)
        else:
            out.append(torch.tensor(head, dtype=odt, device=dev))
        at += count
    if n2 > 0 and (n2 & 1):                                  # exact: it is handed back to NumPy as generator state
        has_gauss, cached = True, _host_gauss(float(xs[n2].item()), float(ar[n2 // 2].item()))
    else:
        has_gauss, cached = False, 0.0
    d.noise, d.position, d.has_gauss, d.gauss, d.patched = out, pos, has_gauss, cached, patched
    return d


RISKY_CAP = 1 << 16


def _normals_fused(raw, pos, has_gauss, gauss, requests, counts, total, n2) -> Draw:
    """`normals` for float32 results on the device in two kernels around one prefix sum (imgxf_np_accept, torch.cumsum,
    imgxf_np_normals_f32) instead of a few dozen elementwise passes; the same arithmetic, statement by statement."""
    from . import _ffi as F
    dev = raw.device
    d = Draw()
    lead = 1 if has_gauss else 0
    out = torch.empty((total,), dtype=torch.float32, device=dev)
    begins, at = [], 0
    for c in counts:
        begins.append(at)
        at += c
    live = [(b, float(s)) for b, c, (_, s) in zip(begins, counts, requests) if c]
    if has_gauss:
        out[0] = float(np.float32(0.0 + live[0][1] * gauss))
    patched = 0
    if n2 > 0:
        groups = (n2 + 1) // 2
        w = words_needed(n2)
        if pos + w > raw.numel():
            raise ValueError("the MT19937 stream is shorter than the draw's margin")
        ng = w // 4
        with torch.cuda.device(dev):
            cs = torch.cuda.current_stream(dev).cuda_stream
            words = raw.data_ptr() + 4 * pos
            acc = torch.empty((ng,), dtype=torch.uint8, device=dev)
            F.call("imgxf_np_accept", words, ng, acc.data_ptr(), cs)
            rank = torch.cumsum(acc, 0, dtype=torch.int64)
            table = np.zeros(len(live), dtype=[("begin", "<i8"), ("scale", "<f8")])
            table["begin"], table["scale"] = [b for b, _ in live], [s for _, s in live]
            reqs_d = torch.from_numpy(table.view(np.uint8).copy()).to(dev)
            info = torch.tensor([-1, 0], dtype=torch.int64, device=dev)
            risky = torch.empty((RISKY_CAP,), dtype=torch.int64, device=dev)
            xr = torch.zeros((2 + 2 * RISKY_CAP,), dtype=torch.float64, device=dev)
            F.call("imgxf_np_normals_f32", words, ng, rank.data_ptr(), groups, n2, lead, reqs_d.data_ptr(), len(live), MARGIN, out.data_ptr(),
                   info.data_ptr(), risky.data_ptr(), RISKY_CAP, xr.data_ptr(), cs)
            last, nrisky = info.cpu().tolist()
        if last < 0:
            raise ValueError("too few accepted groups inside the margin")
        if nrisky > RISKY_CAP:
            raise ValueError("more samples near a float32 rounding boundary than the list holds")
        if nrisky:
            e = risky[:nrisky].cpu().numpy()
            xv = xr[2:2 + 2 * nrisky].cpu().numpy().reshape(-1, 2)
            b_arr = np.array([b for b, _ in live]); s_arr = np.array([s for _, s in live])
            which = np.searchsorted(b_arr, e + lead, side="right") - 1
            fix = np.array([np.float32(0.0 + s_arr[k] * _host_gauss(float(x), float(r))) for k, (x, r) in zip(which, xv)], np.float32)
            out[torch.from_numpy(e + lead).to(dev)] = torch.from_numpy(fix).to(dev)
            patched = int(nrisky)
        pos += 4 * (last + 1)
        if n2 & 1:
            x1, r2 = xr[:2].cpu().tolist()
            has_gauss, cached = True, _host_gauss(x1, r2)
        else:
            has_gauss, cached = False, 0.0
    else:
        has_gauss, cached = False, 0.0
    d.noise = [out[b:b + c] for b, c in zip(begins, counts)]
    d.position, d.has_gauss, d.gauss, d.patched = pos, has_gauss, cached, patched
    return d


def state_at(raw: torch.Tensor, position: int, start: int):
    """(key[624] as a uint32 NumPy array, pos) of the generator after consuming the stream up to `position` — in NumPy's own
    representation: the block is only regenerated by the NEXT request, so a position on a block boundary is `pos = 624`
    of the block before."""
    b, p = divmod(position, 624)
    if p == 0 and position > 0 and position != start:
        b, p = b - 1, 624
    key = (raw[b * 624:(b + 1) * 624].to(torch.int64) & 0xFFFFFFFF).to("cpu").numpy().astype(np.uint32)
    return key, p



# ---- the state sequence, sequentially or in stretches ---------------------------------------------------------------
_JUMP = {"state": None}          # None: not tried yet; False: unavailable / failed its self-check; dict: coefficients on the device


def _jump_tables(device: torch.device):
    """The jump polynomial for a stride of 624 * 2^k words (imagetransformations_amd/mt19937_jump.npz, written and checked on
    the host by tools/make_mt_jump.py) on the device — after a one-time check of the device kernels against the sequential
    generator (one stride: 19 ms)."""
    from . import _ffi as F
    st = _JUMP["state"]
    if st is not None:
        return st if st else None
    _JUMP["state"] = False
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mt19937_jump.npz")
    if not os.path.exists(path) or os.environ.get("IMGXF_MT_JUMP", "1") == "0":
        return None
    z = np.load(path)
    bps = 1 << int(z["log2_blocks"])
    coef = torch.from_numpy(z["coef"].astype(np.uint8).copy()).to(device)
    with torch.cuda.device(device):
        cs = torch.cuda.current_stream(device).cuda_stream
        key = np.random.RandomState(20240229).get_state()[1].astype(np.uint32)
        keys = torch.zeros((3, 624), dtype=torch.int32, device=device)
        keys[0] = torch.from_numpy(key.view(np.int32).copy()).to(device)
        F.call("imgxf_mt19937_jump", keys.data_ptr(), keys[1:].data_ptr(), 2, coef.data_ptr(), cs)
        seq = torch.empty(((2 * bps + 1) * 624,), dtype=torch.int32, device=device)
        F.call("imgxf_mt19937_blocks", keys.data_ptr(), seq.data_ptr(), 2 * bps, cs)
        ok = True
        for m in (1, 2):                                             # (only the top bit of word 0 is state)
            a, b = keys[m].to(torch.int64) & 0xFFFFFFFF, seq[m * bps * 624:(m * bps + 1) * 624].to(torch.int64) & 0xFFFFFFFF
            ok = ok and bool((a[1:] == b[1:]).all()) and int(a[0]) >> 31 == int(b[0]) >> 31
    if ok:
        _JUMP["state"] = {"coef": coef, "bps": bps, "jumps": int(coef.shape[0])}
        return _JUMP["state"]
    return None


def generate_stream(key_d: torch.Tensor, nblocks: int, device: torch.device, cuda_stream: int) -> torch.Tensor:
    """(nblocks + 1) * 624 raw state words from the key (int32 device tensor of 624 words): block 0 = the key.  Long requests
    are cut into stretches of 2^k blocks whose start states come from the jump-ahead kernel (sequential, ~3 ms each) and which
    are then generated by one workgroup each — the block recurrence itself is serial (290 ns per block)."""
    from . import _ffi as F
    total = nblocks + 1
    raw = torch.empty((total * 624,), dtype=torch.int32, device=device)
    jt = _jump_tables(device) if total > (1 << 15) else None
    if jt is None or total <= jt["bps"]:
        F.call("imgxf_mt19937_blocks", key_d.data_ptr(), raw.data_ptr(), nblocks, cuda_stream)
        return raw
    n_st = -(-total // jt["bps"])
    keys = torch.empty((n_st, 624), dtype=torch.int32, device=device)
    keys[0] = key_d
    done = 1                                                 # every jump of a launch starts from the same key and runs in parallel
    while done < n_st:
        k = min(jt["jumps"], n_st - done)
        F.call("imgxf_mt19937_jump", keys[done - 1].data_ptr(), keys[done].data_ptr(), k, jt["coef"].data_ptr(), cuda_stream)
        done += k
    F.call("imgxf_mt19937_stretches", keys.data_ptr(), raw.data_ptr(), n_st, jt["bps"], total, cuda_stream)
    # (word 0 of a jumped key is state only in its top bit: the stream's word at a stretch boundary is written by the stretch
    # before it, one step past its last block)
    return raw

PASS_NORMALS = 1 << 27       # normals per pass over the stream (the pass holds ~70 bytes per normal on the device for a moment)


def draw_on_device(requests: Sequence[tuple], device, f64: bool = False) -> List[torch.Tensor] | None:
    """_draw_pass over stretches of at most PASS_NORMALS normals (np.random's state carries from one to the next)."""
    requests = [(int(n), float(s)) for n, s in requests]
    out: List[torch.Tensor] = []
    i = 0
    while i < len(requests):
        j, tot = i, 0
        while j < len(requests) and (j == i or tot + requests[j][0] <= PASS_NORMALS):
            tot += requests[j][0]
            j += 1
        got = _draw_pass(requests[i:j], device, f64)
        if got is None:
            if i == 0:
                return None
            got = [torch.from_numpy(np.random.normal(0, s, n) if f64 else np.random.normal(0, s, n).astype(np.float32)).to(device)
                   for n, s in requests[i:j]]                    # (only if the generator changed kind in between)
        out.extend(got)
        i = j
    return out


def _draw_pass(requests: Sequence[tuple], device, f64: bool = False) -> List[torch.Tensor] | None:
    """The float32 results of `[np.random.normal(0, scale, count).astype(np.float32) for count, scale in requests]` as
    device tensors, with np.random's global state advanced exactly as those calls would have advanced it — or None (state
    untouched) if the global generator is not the legacy MT19937: the caller then makes the calls on the host."""
    from . import _ffi as F
    device = torch.device(device)
    requests = [(int(n), float(s)) for n, s in requests]
    if not any(n for n, _ in requests):
        return [torch.empty((0,), dtype=torch.float64 if f64 else torch.float32, device=device) for _ in requests]
    kind, key, pos, has_gauss, gauss = np.random.get_state()
    if kind != "MT19937":
        return None
    total = words_needed(sum(n for n, _ in requests))
    nblocks = (int(pos) + total) // 624 + 2
    with torch.cuda.device(device):
        key_d = torch.from_numpy(key.astype(np.uint32).view(np.int32).copy()).to(device)
        raw = generate_stream(key_d, nblocks, device, torch.cuda.current_stream(device).cuda_stream)
        d = normals(raw, int(pos), bool(has_gauss), float(gauss), requests, f64)
        if d.position != int(pos) or bool(has_gauss) != d.has_gauss:
            k, p = state_at(raw, d.position, int(pos))
            np.random.set_state((kind, k, p, int(d.has_gauss), float(d.gauss) if d.has_gauss else 0.0))
    return d.noise


class PendingDraw:
    """draw_on_device in two halves, so that the MT19937 block kernel — one workgroup, 0.17 s for the 144 M normals of 256
    ImageNet-size images — runs on a side stream while the caller queues its other device work and does its host work:
    `PendingDraw(requests, device)` reads np.random's state and launches the kernel, `result()` (later, same thread) evaluates
    the stream, advances np.random and returns the tensors, or None where draw_on_device would.  Nothing else may use
    np.random in between (the state is read at the start and set at the end)."""

    def __init__(self, requests: Sequence[tuple], device, f64: bool = False):
        from . import _ffi as F
        self.requests = [(int(n), float(s)) for n, s in requests]
        self.device, self.f64 = torch.device(device), f64
        self.raw = None
        total = sum(n for n, _ in self.requests)
        self.state = np.random.get_state()
        kind, key, pos = self.state[0], self.state[1], self.state[2]
        if kind != "MT19937" or total == 0 or total > PASS_NORMALS:
            return                                               # result() takes the one-call path
        nblocks = (int(pos) + words_needed(total)) // 624 + 2
        with torch.cuda.device(self.device):
            main = torch.cuda.current_stream(self.device)
            self.side = _side_stream(self.device)
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                self.key_d = torch.from_numpy(key.astype(np.uint32).view(np.int32).copy()).to(self.device)
                self.raw = generate_stream(self.key_d, nblocks, self.device, self.side.cuda_stream)
            self.done = torch.cuda.Event()
            self.done.record(self.side)

    def result(self) -> List[torch.Tensor] | None:
        if self.raw is None:
            return draw_on_device(self.requests, self.device, self.f64)
        kind, key, pos, has_gauss, gauss = self.state
        with torch.cuda.device(self.device):
            main = torch.cuda.current_stream(self.device)
            main.wait_event(self.done)
            self.raw.record_stream(main)
            d = normals(self.raw, int(pos), bool(has_gauss), float(gauss), self.requests, self.f64)
            if d.position != int(pos) or bool(has_gauss) != d.has_gauss:
                k, p = state_at(self.raw, d.position, int(pos))
                np.random.set_state((kind, k, p, int(d.has_gauss), float(d.gauss) if d.has_gauss else 0.0))
        return d.noise


_SIDE: dict = {}


def _side_stream(device: torch.device):
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]
