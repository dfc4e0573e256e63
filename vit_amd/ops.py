"""Tensor-level wrappers over the libvitk C ABI.

Each function checks device/dtype/contiguity, passes raw device pointers and
the current HIP stream to the matching `vitk_*` entry point (include/vitk.h)
and raises on any non-zero return.  torch is used for memory and streams only.
"""
import torch

from . import lib as _lib

_event_hook = None   # optional callable(name) -> context manager, set by bench.py for per-kernel timing


def set_event_hook(hook):
    """Install `hook(name)` (a context-manager factory) around every kernel call, or None."""
    global _event_hook
    _event_hook = hook


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL = _Null()


def _span(name):
    return _event_hook(name) if _event_hook is not None else _NULL


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _scratch_key(device, *extra):
    """Key of a cached scratch buffer: one per device AND stream, so that launches queued on two streams of a
    device never share partial-sum buffers."""
    return (device, torch.cuda.current_stream(device).cuda_stream) + extra


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.VitkError(f"{name} must be a CUDA/HIP tensor: the ViT forward path has no CPU fallback")
    if t.dtype != dtype:
        raise _lib.VitkError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.VitkError(f"{name} must be contiguous")


def _ptr(t):
    return None if t is None else t.data_ptr()


def layernorm(x, gamma, beta, eps, out, rows=None, x_row_stride=None):
    """out[r] = LN(x[r]); x f32 [..., D] (or strided rows), out bf16/f32 [rows, D]."""
    L = _lib.load()
    D = gamma.numel()
    _chk(gamma, torch.float32, "gamma")
    _chk(beta, torch.float32, "beta")
    if x.dtype != torch.float32 or not x.is_cuda:
        raise _lib.VitkError("layernorm: x must be a CUDA float32 tensor")
    if rows is None:
        _chk(x, torch.float32, "x")
        rows = x.numel() // D
        x_row_stride = D
    _chk(out, out.dtype, "out")
    od = _lib.VITK_BF16 if out.dtype == torch.bfloat16 else _lib.VITK_F32
    with _span("layernorm"):
        _lib.check(L.vitk_layernorm(_ptr(x), x_row_stride, _ptr(gamma), _ptr(beta), _ptr(out), D, rows, D,
                                    float(eps), od, _stream()), L)
    return out


def cast_bf16(x, out=None):
    L = _lib.load()
    _chk(x, torch.float32, "x")
    if out is None:
        out = torch.empty_like(x, dtype=torch.bfloat16)
    _chk(out, torch.bfloat16, "out")
    with _span("cast"):
        _lib.check(L.vitk_cast_f32_bf16(_ptr(x), _ptr(out), x.numel(), _stream()), L)
    return out


def patch_embed(img, w, bias, cls, pos, out, patch):
    """img bf16 [B,3,H,W]; w bf16 [D,3*P*P]; out f32 [B,ntok,D]."""
    L = _lib.load()
    _chk(img, torch.bfloat16, "img")
    _chk(w, torch.bfloat16, "w")
    for n, t in (("bias", bias), ("cls", cls), ("pos", pos), ("out", out)):
        _chk(t, torch.float32, n)
    B, C, H, W = img.shape
    D = w.shape[0]
    if C != 3 or w.shape[1] != 3 * patch * patch:
        raise _lib.VitkError(f"patch_embed: weight {tuple(w.shape)} does not match 3x{patch}x{patch} patches")
    ntok = (H // patch) * (W // patch) + 1
    if tuple(out.shape) != (B, ntok, D) or pos.numel() != ntok * D:
        raise _lib.VitkError(f"patch_embed: out {tuple(out.shape)} / pos {tuple(pos.shape)} do not match "
                             f"[{B},{ntok},{D}]")
    with _span("patch_embed"):
        _lib.check(L.vitk_patch_embed(_ptr(img), _ptr(w), _ptr(bias), _ptr(cls), _ptr(pos), _ptr(out),
                                      B, H, W, patch, D, _stream()), L)
    return out


def patch_embed_cols(img, w_padded, bias, cls, pos, out, cols, patch):
    """Any patch size: img bf16 [B,3,H,W]; w_padded bf16 [D,Kpad] (Kpad = 3*P*P rounded up to 64, zero
    padded); cols bf16 [B*npatch, Kpad] scratch; out f32 [B,ntok,D]."""
    L = _lib.load()
    _chk(img, torch.bfloat16, "img"); _chk(w_padded, torch.bfloat16, "w_padded"); _chk(cols, torch.bfloat16, "cols")
    for n, t in (("bias", bias), ("cls", cls), ("pos", pos), ("out", out)):
        _chk(t, torch.float32, n)
    B, C, H, W = img.shape
    D = w_padded.shape[0]
    kpad = (3 * patch * patch + 63) // 64 * 64
    npatch = (H // patch) * (W // patch)
    if C != 3 or w_padded.shape[1] != kpad or tuple(cols.shape) != (B * npatch, kpad):
        raise _lib.VitkError(f"patch_embed_cols: weight {tuple(w_padded.shape)} / cols {tuple(cols.shape)} do not "
                             f"match [{D},{kpad}] / [{B * npatch},{kpad}]")
    if tuple(out.shape) != (B, npatch + 1, D) or pos.numel() != (npatch + 1) * D:
        raise _lib.VitkError(f"patch_embed_cols: out {tuple(out.shape)} / pos {tuple(pos.shape)} do not match "
                             f"[{B},{npatch + 1},{D}]")
    with _span("patch_embed"):
        _lib.check(L.vitk_patch_embed_cols(_ptr(img), _ptr(w_padded), _ptr(bias), _ptr(cls), _ptr(pos), _ptr(out),
                                           _ptr(cols), B, H, W, patch, D, _stream()), L)
    return out


def _dense_shapes(a, w, out, name):
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K or tuple(out.shape) != (M, N):
        raise _lib.VitkError(f"{name}: shapes A{tuple(a.shape)} W{tuple(w.shape)} out{tuple(out.shape)}")
    return M, N, K


def gemm_bias(a, w, bias, out):
    L = _lib.load()
    _chk(a, torch.bfloat16, "A"); _chk(w, torch.bfloat16, "W"); _chk(bias, torch.float32, "bias")
    _chk(out, out.dtype, "out")
    M, N, K = _dense_shapes(a, w, out, "gemm_bias")
    od = _lib.VITK_BF16 if out.dtype == torch.bfloat16 else _lib.VITK_F32
    with _span("gemm_bias"):
        _lib.check(L.vitk_gemm_bias(_ptr(a), _ptr(w), _ptr(bias), _ptr(out), M, N, K, od, _stream()), L)
    return out


def gemm_bias_gelu(a, w, bias, out):
    L = _lib.load()
    _chk(a, torch.bfloat16, "A"); _chk(w, torch.bfloat16, "W"); _chk(bias, torch.float32, "bias")
    _chk(out, torch.bfloat16, "out")
    M, N, K = _dense_shapes(a, w, out, "gemm_bias_gelu")
    with _span("gemm_bias_gelu"):
        _lib.check(L.vitk_gemm_bias_gelu(_ptr(a), _ptr(w), _ptr(bias), _ptr(out), M, N, K, _stream()), L)
    return out


def gemm_bias_tanh(a, w, bias, out):
    L = _lib.load()
    _chk(a, torch.bfloat16, "A"); _chk(w, torch.bfloat16, "W"); _chk(bias, torch.float32, "bias")
    _chk(out, torch.bfloat16, "out")
    M, N, K = _dense_shapes(a, w, out, "gemm_bias_tanh")
    with _span("gemm_bias_tanh"):
        _lib.check(L.vitk_gemm_bias_tanh(_ptr(a), _ptr(w), _ptr(bias), _ptr(out), M, N, K, _stream()), L)
    return out


def gemm_bias_residual(a, w, bias, resid, out):
    """out f32 = resid + a @ w.T + bias; `out` may be `resid` itself."""
    L = _lib.load()
    _chk(a, torch.bfloat16, "A"); _chk(w, torch.bfloat16, "W"); _chk(bias, torch.float32, "bias")
    _chk(resid, torch.float32, "resid"); _chk(out, torch.float32, "out")
    M, N, K = _dense_shapes(a, w, out, "gemm_bias_residual")
    if tuple(resid.shape) != (M, N):
        raise _lib.VitkError(f"gemm_bias_residual: resid {tuple(resid.shape)} vs out [{M},{N}]")
    with _span("gemm_bias_residual"):
        _lib.check(L.vitk_gemm_bias_residual(_ptr(a), _ptr(w), _ptr(bias), _ptr(resid), _ptr(out), M, N, K,
                                             _stream()), L)
    return out


def ln_fold_ok(rows, D, mlp):
    """Shapes for which the LayerNorm between a residual GEMM and the following projection can be folded into the
    two (vitk_gemm_bias_residual_ln / vitk_gemm_ln_bias_gelu / vitk_qkv_proj_ln): the two-group persistent GEMM's
    shapes with every tensor below 4 GiB."""
    return (rows >= 128 and D % 256 == 0 and mlp % 256 == 0 and D % 128 == 0 and mlp % 128 == 0
            and (rows + 256) * max(3 * D, mlp) * 4 < (1 << 32))


def gemm_bias_residual_ln(a, w, bias, resid, out, out16, stats):
    """out f32 = resid + a @ w.T + bias (may be `resid` itself); out16 = bf16(out); stats f32 [M, N/64, 2] = per row
    and 64-column strip (sum, sum of squares) of the new row."""
    L = _lib.load()
    _chk(a, torch.bfloat16, "A"); _chk(w, torch.bfloat16, "W"); _chk(bias, torch.float32, "bias")
    _chk(resid, torch.float32, "resid"); _chk(out, torch.float32, "out"); _chk(out16, torch.bfloat16, "out16")
    _chk(stats, torch.float32, "stats")
    M, N, K = _dense_shapes(a, w, out, "gemm_bias_residual_ln")
    if tuple(resid.shape) != (M, N) or tuple(out16.shape) != (M, N) or stats.numel() != M * (N // 64) * 2:
        raise _lib.VitkError(f"gemm_bias_residual_ln: resid {tuple(resid.shape)} out16 {tuple(out16.shape)} "
                             f"stats {tuple(stats.shape)} vs out [{M},{N}]")
    with _span("gemm_bias_residual"):
        _lib.check(L.vitk_gemm_bias_residual_ln(_ptr(a), _ptr(w), _ptr(bias), _ptr(resid), _ptr(out), _ptr(out16),
                                                _ptr(stats), M, N, K, _stream()), L)
    return out


def gemm_bias_residual_planes(a, w, bias, plane_h, plane_l, stats=None, x_in=None, x_out=None):
    """The residual GEMM x' = x + a @ w.T + bias with the stream x held as two 16-bit planes (plane_h = bf16(x) rounded
    half up, plane_l = the lower half of x's f32 bit pattern; see planes_to_f32).  x_in (f32): read x from it and write
    the planes; x_out (f32): read the planes and write x' there; neither: planes in place.  stats f32 [M, N/64, 2]
    receives the new rows' strip statistics whenever the planes are written."""
    L = _lib.load()
    _chk(a, torch.bfloat16, "A"); _chk(w, torch.bfloat16, "W"); _chk(bias, torch.float32, "bias")
    _chk(plane_h, torch.bfloat16, "plane_h"); _chk(plane_l, torch.bfloat16, "plane_l")
    M, N, K = _dense_shapes(a, w, plane_h, "gemm_bias_residual_planes")
    if tuple(plane_l.shape) != (M, N):
        raise _lib.VitkError(f"gemm_bias_residual_planes: plane_l {tuple(plane_l.shape)} vs [{M},{N}]")
    for t, n in ((x_in, "x_in"), (x_out, "x_out")):
        if t is not None:
            _chk(t, torch.float32, n)
            if tuple(t.shape) != (M, N):
                raise _lib.VitkError(f"gemm_bias_residual_planes: {n} {tuple(t.shape)} vs [{M},{N}]")
    if stats is not None:
        _chk(stats, torch.float32, "stats")
        if stats.numel() != M * (N // 64) * 2:
            raise _lib.VitkError(f"gemm_bias_residual_planes: stats {tuple(stats.shape)} vs [{M},{N // 64},2]")
    with _span("gemm_bias_residual"):
        _lib.check(L.vitk_gemm_bias_residual_planes(_ptr(a), _ptr(w), _ptr(bias), _ptr(x_in) if x_in is not None else None,
                                                    _ptr(plane_h), _ptr(plane_l), _ptr(x_out) if x_out is not None else None,
                                                    _ptr(stats) if stats is not None else None, M, N, K, _stream()), L)
    return x_out if x_out is not None else plane_h


def planes_to_f32(plane_h, plane_l):
    """The f32 tensor two 16-bit planes stand for: bits(x) = (plane_h << 16) + sext16(plane_l), exactly."""
    h = plane_h.view(torch.int16).to(torch.int32)
    lo = plane_l.view(torch.int16).to(torch.int32)
    return ((h << 16) + lo).view(torch.float32)


def f32_to_planes(x):
    """(plane_h, plane_l) of an f32 tensor: plane_h = upper half of bits(x) + 0x8000 (bf16 rounded half up), plane_l =
    lower half of bits(x).  torch ops; the GEMM epilogues write the same."""
    b = x.contiguous().view(torch.int32)
    h = ((b + 0x8000) >> 16).to(torch.int16).view(torch.bfloat16)
    lo = (b << 16 >> 16).to(torch.int16).view(torch.bfloat16)
    return h, lo


def ln_finalize(stats, ab, D, eps):
    """ab f32 [rows, 2] = (rstd, -mean * rstd) per row from the strip statistics of a D-column LayerNorm."""
    L = _lib.load()
    _chk(stats, torch.float32, "stats"); _chk(ab, torch.float32, "ab")
    rows = ab.shape[0]
    if ab.numel() != rows * 2 or stats.numel() != rows * (D // 64) * 2:
        raise _lib.VitkError(f"ln_finalize: stats {tuple(stats.shape)} ab {tuple(ab.shape)} D={D}")
    with _span("ln_finalize"):
        _lib.check(L.vitk_ln_finalize(_ptr(stats), _ptr(ab), rows, D, float(eps), _stream()), L)
    return ab


def gemm_ln_bias_gelu(a, wg, ab, colsum, c, out):
    """out bf16 = gelu(LN(x) @ w.T + bias) from a = bf16(x), wg = bf16(w * gamma), the row factors ab and the
    column vectors colsum[n] = sum_k wg[n, k], c[n] = bias[n] + sum_k beta[k] w[n, k]."""
    L = _lib.load()
    _chk(a, torch.bfloat16, "A"); _chk(wg, torch.bfloat16, "Wg"); _chk(out, torch.bfloat16, "out")
    for n, t in (("ab", ab), ("colsum", colsum), ("c", c)):
        _chk(t, torch.float32, n)
    M, N, K = _dense_shapes(a, wg, out, "gemm_ln_bias_gelu")
    if ab.numel() != 2 * M or colsum.numel() != N or c.numel() != N:
        raise _lib.VitkError("gemm_ln_bias_gelu: ab / colsum / c sizes")
    with _span("gemm_bias_gelu"):
        _lib.check(L.vitk_gemm_ln_bias_gelu(_ptr(a), _ptr(wg), _ptr(ab), _ptr(colsum), _ptr(c), _ptr(out), M, N, K,
                                            _stream()), L)
    return out


def qkv_proj_ln(a, wg, ab, colsum, c, qkv, batch, ntok, heads, hd):
    """qkv_proj of LN(x) from a = bf16(x) [batch*ntok, D] (see gemm_ln_bias_gelu for the other operands)."""
    L = _lib.load()
    _chk(a, torch.bfloat16, "A"); _chk(wg, torch.bfloat16, "Wg"); _chk(qkv, torch.bfloat16, "qkv")
    for n, t in (("ab", ab), ("colsum", colsum), ("c", c)):
        _chk(t, torch.float32, n)
    D = heads * hd
    if tuple(a.shape) != (batch * ntok, D) or tuple(wg.shape) != (3 * D, D) or \
            tuple(qkv.shape) != (3, batch, heads, ntok, hd) or ab.numel() != 2 * batch * ntok or \
            colsum.numel() != 3 * D or c.numel() != 3 * D:
        raise _lib.VitkError(f"qkv_proj_ln: shapes A{tuple(a.shape)} W{tuple(wg.shape)} qkv{tuple(qkv.shape)}")
    with _span("qkv_proj"):
        _lib.check(L.vitk_qkv_proj_ln(_ptr(a), _ptr(wg), _ptr(ab), _ptr(colsum), _ptr(c), _ptr(qkv), batch, ntok,
                                      heads, hd, _stream()), L)
    return qkv


def qkv_proj(a, w, bias, qkv, batch, ntok, heads, hd):
    """a bf16 [batch*ntok, D]; qkv bf16 [3, batch, heads, ntok, hd]."""
    L = _lib.load()
    _chk(a, torch.bfloat16, "A"); _chk(w, torch.bfloat16, "W"); _chk(bias, torch.float32, "bias")
    _chk(qkv, torch.bfloat16, "qkv")
    D = heads * hd
    if tuple(a.shape) != (batch * ntok, D) or tuple(w.shape) != (3 * D, D) or \
            tuple(qkv.shape) != (3, batch, heads, ntok, hd):
        raise _lib.VitkError(f"qkv_proj: shapes A{tuple(a.shape)} W{tuple(w.shape)} qkv{tuple(qkv.shape)}")
    with _span("qkv_proj"):
        _lib.check(L.vitk_qkv_proj(_ptr(a), _ptr(w), _ptr(bias), _ptr(qkv), batch, ntok, heads, hd, _stream()), L)
    return qkv


def attention(qkv, ctx, scale, rows=None):
    """qkv bf16 [3, B, H, N, hd] -> ctx bf16 [B, N, H*hd]; with `rows` only the query rows below it are computed
    (whole 32-row tiles; the rest of ctx is left as it was)."""
    L = _lib.load()
    _chk(qkv, torch.bfloat16, "qkv"); _chk(ctx, torch.bfloat16, "ctx")
    _, B, H, N, hd = qkv.shape
    if tuple(ctx.shape) != (B, N, H * hd):
        raise _lib.VitkError(f"attention: ctx {tuple(ctx.shape)} vs [{B},{N},{H * hd}]")
    with _span("attention"):
        if rows is None:
            _lib.check(L.vitk_attention(_ptr(qkv[0]), _ptr(qkv[1]), _ptr(qkv[2]), _ptr(ctx), B, H, N, hd,
                                        float(scale), _stream()), L)
        else:
            _lib.check(L.vitk_attention_rows(_ptr(qkv[0]), _ptr(qkv[1]), _ptr(qkv[2]), _ptr(ctx), B, H, N, hd,
                                             float(scale), int(rows), _stream()), L)
    return ctx


def gemm_bias_residual_rows(a, lda, w, bias, x, ldo, rows):
    """x[m*ldo : m*ldo + N] += a[m*lda : m*lda + K] @ w.T + bias for m < rows: the residual update on every ldo-th
    row of the f32 stream x (in place) from every lda-th row of the bf16 tensor a."""
    L = _lib.load()
    _chk(a, torch.bfloat16, "A"); _chk(w, torch.bfloat16, "W"); _chk(bias, torch.float32, "bias"); _chk(x, torch.float32, "x")
    N, K = w.shape
    if a.numel() < (rows - 1) * lda + K or x.numel() < (rows - 1) * ldo + N:
        raise _lib.VitkError(f"gemm_bias_residual_rows: {rows} rows at strides {lda} / {ldo} do not fit A {tuple(a.shape)} / x {tuple(x.shape)}")
    with _span("gemm_bias_residual"):
        _lib.check(L.vitk_gemm_bias_residual_strided(_ptr(a), lda, _ptr(w), _ptr(bias), _ptr(x), _ptr(x), ldo, rows, N, K,
                                                     _stream()), L)
    return x


# ---- fp32 compute mode (vitk_*_f32) ------------------------------------------------------------

def gemm_f32(a, w, bias, out, epilogue=_lib.EPI_BIAS, resid=None):
    """out f32 [M,N] = epilogue(a @ w.T + bias); a f32 [M,K], w f32 [N,K]."""
    L = _lib.load()
    for n, t in (("A", a), ("W", w), ("bias", bias), ("resid", resid), ("out", out)):
        _chk(t, torch.float32, n)
    M, N, K = _dense_shapes(a, w, out, "gemm_f32")
    if resid is not None and tuple(resid.shape) != (M, N):
        raise _lib.VitkError(f"gemm_f32: resid {tuple(resid.shape)} vs out [{M},{N}]")
    with _span("gemm_f32"):
        _lib.check(L.vitk_gemm_f32(_ptr(a), _ptr(w), _ptr(bias), _ptr(resid), _ptr(out), M, N, K, int(epilogue),
                                   _stream()), L)
    return out


def qkv_proj_f32(a, w, bias, qkv, batch, ntok, heads, hd):
    """a f32 [batch*ntok, D]; qkv f32 [3, batch, heads, ntok, hd]."""
    L = _lib.load()
    for n, t in (("A", a), ("W", w), ("bias", bias), ("qkv", qkv)):
        _chk(t, torch.float32, n)
    D = heads * hd
    if tuple(a.shape) != (batch * ntok, D) or tuple(w.shape) != (3 * D, D) or \
            tuple(qkv.shape) != (3, batch, heads, ntok, hd):
        raise _lib.VitkError(f"qkv_proj_f32: shapes A{tuple(a.shape)} W{tuple(w.shape)} qkv{tuple(qkv.shape)}")
    with _span("qkv_proj_f32"):
        _lib.check(L.vitk_qkv_proj_f32(_ptr(a), _ptr(w), _ptr(bias), _ptr(qkv), batch, ntok, heads, hd, _stream()), L)
    return qkv


def patch_embed_f32(img, w, bias, cls, pos, out, patch):
    """img f32 [B,3,H,W]; w f32 [D,3*P*P]; out f32 [B,ntok,D]."""
    L = _lib.load()
    for n, t in (("img", img), ("w", w), ("bias", bias), ("cls", cls), ("pos", pos), ("out", out)):
        _chk(t, torch.float32, n)
    B, C, H, W = img.shape
    D = w.shape[0]
    if C != 3 or w.shape[1] != 3 * patch * patch:
        raise _lib.VitkError(f"patch_embed_f32: weight {tuple(w.shape)} does not match 3x{patch}x{patch} patches")
    ntok = (H // patch) * (W // patch) + 1
    if tuple(out.shape) != (B, ntok, D) or pos.numel() != ntok * D:
        raise _lib.VitkError(f"patch_embed_f32: out {tuple(out.shape)} / pos {tuple(pos.shape)} do not match "
                             f"[{B},{ntok},{D}]")
    with _span("patch_embed_f32"):
        _lib.check(L.vitk_patch_embed_f32(_ptr(img), _ptr(w), _ptr(bias), _ptr(cls), _ptr(pos), _ptr(out),
                                          B, H, W, patch, D, _stream()), L)
    return out


def attention_f32(qkv, ctx, scale):
    """qkv f32 [3, B, H, N, hd] -> ctx f32 [B, N, H*hd]."""
    L = _lib.load()
    _chk(qkv, torch.float32, "qkv"); _chk(ctx, torch.float32, "ctx")
    _, B, H, N, hd = qkv.shape
    if tuple(ctx.shape) != (B, N, H * hd):
        raise _lib.VitkError(f"attention_f32: ctx {tuple(ctx.shape)} vs [{B},{N},{H * hd}]")
    with _span("attention_f32"):
        _lib.check(L.vitk_attention_f32(_ptr(qkv[0]), _ptr(qkv[1]), _ptr(qkv[2]), _ptr(ctx), B, H, N, hd,
                                        float(scale), _stream()), L)
    return ctx


def normalize_u8(img, mean, std, out):
    """out = (float(img) - mean[c]) / std[c]; img uint8 [B,3,H,W], out bf16 or f32 of the same shape."""
    L = _lib.load()
    _chk(img, torch.uint8, "img")
    _chk(out, out.dtype, "out")
    if img.dim() != 4 or img.shape[1] != 3 or out.shape != img.shape:
        raise _lib.VitkError(f"normalize_u8: expected [B,3,H,W] in and out, got {tuple(img.shape)} / {tuple(out.shape)}")
    if out.dtype not in (torch.bfloat16, torch.float32):
        raise _lib.VitkError(f"normalize_u8: out must be bf16 or f32, got {out.dtype}")
    od = _lib.VITK_BF16 if out.dtype == torch.bfloat16 else _lib.VITK_F32
    m = (_lib.ctypes.c_float * 3)(*[float(v) for v in mean])
    sd = (_lib.ctypes.c_float * 3)(*[float(v) for v in std])
    B, _, H, W = img.shape
    with _span("normalize_u8"):
        _lib.check(L.vitk_normalize_u8(_ptr(img), _ptr(out), B, H, W, m, sd, od, _stream()), L)
    return out


def target_rank(logits, target, out=None):
    """rank[r] = number of classes placed above target[r] in logits[r] (int32 [rows])."""
    L = _lib.load()
    _chk(logits, torch.float32, "logits")
    _chk(target, torch.int64, "target")
    if logits.dim() != 2 or target.dim() != 1 or target.shape[0] != logits.shape[0]:
        raise _lib.VitkError(f"target_rank: expected [rows,classes] and [rows], got {tuple(logits.shape)} / {tuple(target.shape)}")
    if out is None:
        out = torch.empty(logits.shape[0], dtype=torch.int32, device=logits.device)
    _chk(out, torch.int32, "rank")
    with _span("target_rank"):
        _lib.check(L.vitk_target_rank(_ptr(logits), _ptr(target), _ptr(out), logits.shape[0], logits.shape[1],
                                      _stream()), L)
    return out


_ln_partials = {}


def layernorm_backward(x, gamma, g, eps, add=None, dx_bf16=None):
    """(dx f32 [rows, D], dgamma f32 [D], dbeta f32 [D]) of y = LayerNorm(x) for the output gradient g
    (bf16 or f32 [rows, D]); x f32 [rows, D] contiguous; add (f32 [rows, D], optional) is added to dx;
    dx_bf16 (bf16 [rows, D], optional) receives dx rounded to bf16."""
    L = _lib.load()
    _chk(x, torch.float32, "x")
    _chk(gamma, torch.float32, "gamma")
    _chk(g, g.dtype, "g")
    _chk(add, torch.float32, "add")
    if add is not None and add.shape != x.shape:
        raise _lib.VitkError(f"layernorm_backward: add {tuple(add.shape)} vs x {tuple(x.shape)}")
    if g.dtype not in (torch.bfloat16, torch.float32) or g.shape != x.shape or x.dim() != 2:
        raise _lib.VitkError(f"layernorm_backward: x {tuple(x.shape)} {x.dtype}, g {tuple(g.shape)} {g.dtype}")
    rows, D = x.shape
    key = _scratch_key(x.device, D)
    part = _ln_partials.get(key)
    if part is None:
        part = _ln_partials[key] = torch.empty(1024, 2, D, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dgamma = torch.empty(D, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(D, dtype=torch.float32, device=x.device)
    gd = _lib.VITK_BF16 if g.dtype == torch.bfloat16 else _lib.VITK_F32
    with _span("layernorm_backward"):
        _lib.check(L.vitk_layernorm_backward(_ptr(x), D, _ptr(gamma), _ptr(g), gd, _ptr(add), _ptr(dx), _ptr(dx_bf16), _ptr(dgamma), _ptr(dbeta),
                                             _ptr(part), part.shape[0], rows, D, float(eps), _stream()), L)
    return dx, dgamma, dbeta


_tn_workspace = {}


def gemm_tn_supported(M, N, R):
    return _lib.load().vitk_gemm_tn_workspace(M, N, R) >= 0


def gemm_tn(g, x, out=None, colsum=None):
    """out f32 [M, N] = g^T x for g bf16 [R, M], x bf16 [R, N] (the weight gradient of a Linear layer).  With
    `colsum` (f32 [M]) the same launch also writes the column sums of g (the layer's bias gradient)."""
    L = _lib.load()
    _chk(g, torch.bfloat16, "g")
    _chk(x, torch.bfloat16, "x")
    if g.dim() != 2 or x.dim() != 2 or g.shape[0] != x.shape[0]:
        raise _lib.VitkError(f"gemm_tn: g {tuple(g.shape)}, x {tuple(x.shape)}")
    R, M = g.shape
    N = x.shape[1]
    need = (L.vitk_gemm_tn_workspace if colsum is None else L.vitk_gemm_tn_colsum_workspace)(M, N, R)
    if need < 0:
        raise _lib.VitkError(f"gemm_tn: unsupported shape M={M} N={N} R={R} (M, N multiples of 256)")
    key = _scratch_key(g.device)
    ws = _tn_workspace.get(key)
    if need > 0 and (ws is None or ws.numel() < need):
        ws = _tn_workspace[key] = torch.empty(need, dtype=torch.float32, device=g.device)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=g.device)
    _chk(out, torch.float32, "out")
    wsp, wsn = (_ptr(ws), ws.numel()) if need > 0 else (None, 0)
    with _span("gemm_tn"):
        if colsum is None:
            _lib.check(L.vitk_gemm_tn(_ptr(g), _ptr(x), _ptr(out), wsp, wsn, M, N, R, _stream()), L)
        else:
            _chk(colsum, torch.float32, "colsum")
            if colsum.numel() != M:
                raise _lib.VitkError(f"gemm_tn: colsum has {colsum.numel()} elements, expected {M}")
            _lib.check(L.vitk_gemm_tn_colsum(_ptr(g), _ptr(x), _ptr(out), _ptr(colsum), wsp, wsn, M, N, R, _stream()), L)
    return out


def attention_lse(qkv, ctx, scale):
    """attention() that also returns the per-row log-sum-exp (f32 [B, H, N]) the backward kernel needs."""
    L = _lib.load()
    _chk(qkv, torch.bfloat16, "qkv")
    _chk(ctx, torch.bfloat16, "ctx")
    _, B, H, N, hd = qkv.shape
    lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    with _span("attention"):
        _lib.check(L.vitk_attention_lse(_ptr(qkv[0]), _ptr(qkv[1]), _ptr(qkv[2]), _ptr(ctx), _ptr(lse), B, H, N, hd,
                                        float(scale), _stream()), L)
    return lse


def attention_backward_supported(N, hd):
    return (hd == 64 and N <= 608) or (hd in (80, 96) and N <= 288)


def attention_backward(qkv, ctx, dctx, lse, scale):
    """Gradient of attention() w.r.t. q, k, v as one bf16 [B, N, 3, H, hd] tensor."""
    L = _lib.load()
    for t, n in ((qkv, "qkv"), (ctx, "ctx"), (dctx, "dctx")):
        _chk(t, torch.bfloat16, n)
    _chk(lse, torch.float32, "lse")
    _, B, H, N, hd = qkv.shape
    if ctx.shape != (B, N, H * hd) or dctx.shape != ctx.shape or lse.shape != (B, H, N):
        raise _lib.VitkError(f"attention_backward: ctx {tuple(ctx.shape)}, dctx {tuple(dctx.shape)}, lse {tuple(lse.shape)}")
    dbuf = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    dqkv = torch.empty(B, N, 3, H, hd, dtype=torch.bfloat16, device=qkv.device)
    with _span("attention_backward"):
        _lib.check(L.vitk_attention_backward(_ptr(qkv[0]), _ptr(qkv[1]), _ptr(qkv[2]), _ptr(ctx), _ptr(dctx), _ptr(lse),
                                             _ptr(dbuf), _ptr(dqkv), B, H, N, hd, float(scale), _stream()), L)
    return dqkv


def gemm_bias_gelu_dual_supported(M, N, K):
    return N % 256 == 0 and K % 64 == 0 and M >= 128


def gemm_bias_gelu_dual(a, w, bias, pre, out):
    """out = gelu(a w^T + bias) and pre = a w^T + bias (both bf16 [M, N])."""
    L = _lib.load()
    M, N, K = _dense_shapes(a, w, out, "gemm_bias_gelu_dual")
    _chk(pre, torch.bfloat16, "pre")
    _chk(out, torch.bfloat16, "out")
    if pre.shape != out.shape:
        raise _lib.VitkError("gemm_bias_gelu_dual: pre and out shapes differ")
    with _span("gemm_bias_gelu"):
        _lib.check(L.vitk_gemm_bias_gelu_dual(_ptr(a), _ptr(w), _ptr(bias), _ptr(pre), _ptr(out), M, N, K, _stream()), L)
    return out


def gemm_grad_act(g, wt, aux, out, act):
    """out bf16 [M, N] = bf16(g @ wt.T) * f'(aux): the input gradient of a Linear layer through the activation in
    front of it.  g bf16 [M, K], wt bf16 [N, K] (the layer's weight transposed), aux bf16 [M, N]: the GELU's
    pre-activation (act = lib.ACT_GELU) or the tanh output (act = lib.ACT_TANH)."""
    L = _lib.load()
    for n, t in (("g", g), ("wt", wt), ("aux", aux), ("out", out)):
        _chk(t, torch.bfloat16, n)
    M, N, K = _dense_shapes(g, wt, out, "gemm_grad_act")
    if tuple(aux.shape) != (M, N):
        raise _lib.VitkError(f"gemm_grad_act: aux {tuple(aux.shape)} vs out [{M},{N}]")
    with _span("gemm_grad_act"):
        _lib.check(L.vitk_gemm_grad_act(_ptr(g), _ptr(wt), _ptr(aux), _ptr(out), M, N, K, int(act), _stream()), L)
    return out


_colsum_partials = {}


def colsum(g):
    """f32 [M] column sums of g bf16 [R, M] (the bias gradient of a Linear layer); M a multiple of 256."""
    L = _lib.load()
    _chk(g, torch.bfloat16, "g")
    R, M = g.shape
    key = _scratch_key(g.device)
    part = _colsum_partials.get(key)
    if part is None or part.numel() < 256 * M:
        part = _colsum_partials[key] = torch.empty(256 * max(M, 4096), dtype=torch.float32, device=g.device)
    out = torch.empty(M, dtype=torch.float32, device=g.device)
    with _span("colsum"):
        _lib.check(L.vitk_colsum_bf16(_ptr(g), _ptr(out), _ptr(part), 256, R, M, _stream()), L)
    return out


def colsum_static(g, out=None):
    """colsum() through vitk_colsum_bf16_static: the partial sums live in the library's built-in device
    buffer, so the call allocates nothing when `out` is given (M <= 4096)."""
    L = _lib.load()
    _chk(g, torch.bfloat16, "g")
    R, M = g.shape
    if out is None:
        out = torch.empty(M, dtype=torch.float32, device=g.device)
    _chk(out, torch.float32, "out")
    if out.numel() != M:
        raise _lib.VitkError(f"colsum_static: out has {out.numel()} elements, expected {M}")
    with torch.cuda.device(g.device), _span("colsum"):
        _lib.check(L.vitk_colsum_bf16_static(_ptr(g), _ptr(out), R, M, _stream()), L)
    return out
