"""CPU oracle for the dots.ocr hot path: vision tower -> merger -> Qwen2 LM -> greedy decode.

TEST INFRASTRUCTURE ONLY.  Nothing under dots_ocr_amd/ or dots_ocr/ imports this module; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the checker.

What it restates (plain PyTorch on CPU, functional over an HF-named state dict):
  * LM: transformers Qwen2 — RMSNorm modeling_qwen2.py:238-252, rotary
    :91-135, attention :176-235, MLP :35-48, decoder layer :258-299, greedy loop of
    GenerationMixin.generate as called at reference dots_ocr/parser.py:110.
    PINNED: tests/test_oracle_pins.py checks logits and greedy tokens against the in-container
    ``transformers.Qwen2ForCausalLM`` (the reference's own dependency, requirements.txt:6).
  * Vision tower: the HF-hub remote code ``modeling_dots_vision.py`` loaded with
    trust_remote_code at reference dots_ocr/parser.py:68-74.  That file is NOT in /root/reference
    and there is no network, so its structure is restated from SURVEY §8(a) [RECALLED]:
    Conv2d(k=s=14) patch embed + RMSNorm, 42 x [RMSNorm -> qkv -> 2-D rope -> bidirectional
    var-len attention -> proj -> +res -> RMSNorm -> SwiGLU(fc1,fc3 -> fc2) -> +res], post-trunk
    RMSNorm, PatchMerger(LayerNorm -> Linear -> GELU -> Linear).
    PARITY UNPINNED for the tower as a whole (no golden vectors exist anywhere in the reference).
    Its building blocks ARE pinned against the in-container transformers analogues: 2-D rope
    position ids (vision_utils.py:81-127), VisionRotaryEmbedding + apply_rotary_pos_emb_vision
    (modeling_qwen2_vl.py:225-248), PatchMerger (modeling_qwen2_vl.py:277-290), and one whole
    transformer BLOCK (norm order, residual placement, qkv split, rope, per-image attention, SwiGLU
    wiring) against Qwen2_5_VLVisionBlock (modeling_qwen2_5_vl.py) on shared weights.

Two numeric modes:
  emulate_bf16=False  fp32 activations (bf16-valued weights): the truth for logit tolerances.
  emulate_bf16=True   rounds to bf16 at exactly the points where the HIP engine stores bf16
                      (DESIGN.md "Numerics contract"); greedy-token equality is asserted
                      against this mode.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _r(x: torch.Tensor, emu: bool) -> torch.Tensor:
    """bf16 storage point."""
    return x.to(torch.bfloat16).float() if emu else x


def _w(sd: SD, name: str) -> torch.Tensor:
    return sd[name].float()


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float, emu: bool) -> torch.Tensor:
    # modeling_qwen2.py:246-252: fp32 normalise, cast to input dtype, then multiply by weight
    xn = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return _r(_r(xn, emu) * w, emu)


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, emu: bool) -> torch.Tensor:
    return _r(F.layer_norm(x, (x.shape[-1],), w, b, eps), emu)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None, a8: bool = False) -> torch.Tensor:
    """a8: the input activations are quantised per row (token) to e4m3 first — what the fp8 engine's ViT / prefill GEMMs do."""
    if a8:
        q, s = quantize_rows_fp8(x)
        x = q * s[:, None]
    y = x @ w.t()
    return y if b is None else y + b


# =============================================================================== fp8 weights (BASELINE configs[4])
def quantize_rows_fp8(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[N, K] -> (q, scale): OCP e4m3 ("e4m3fn") values q (as fp32) and one fp32 scale per output channel,
    scale = max|row| / 448 (1 for an all-zero row), q = e4m3(w / scale) with torch's round-to-nearest-even cast.
    The definition the engine's csrc/quant.hip is tested against (tests/test_fp8_gpu.py)."""
    w = w.float()
    amax = w.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    q = (w / scale[:, None]).to(torch.float8_e4m3fn).float()
    return q, scale


FP8_LINEAR_SUFFIXES = ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight", "mlp.fc3.weight",
                       "merger.mlp.0.weight", "merger.mlp.2.weight",
                       "self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.o_proj.weight",
                       "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight", "lm_head.weight")


def quantize_fp8_state_dict(sd: SD) -> SD:
    """The state dict an fp8_weights engine computes with: every ViT-block / merger / LM linear and the lm_head (a copy of the
    embedding table when tied) replaced by q * scale in fp32; embeddings, the patch embedding, norms and biases untouched.
    The fp8 oracle mode = the oracle on this state dict with generate(..., fp8_act=True) (per-token e4m3 activations into the
    ViT / prefill linears, as the engine's fp8-MFMA GEMMs get them): y = (qa sa) @ (qw sw)^T differs from the engine's
    (qa @ qw^T) sa sw by fp32 rounding only, and the bf16 storage points of emulate_bf16 are those of the bf16 configuration."""
    out = dict(sd)
    if "lm_head.weight" not in out:
        out["lm_head.weight"] = sd["model.embed_tokens.weight"]
    for name in list(out):
        if name.endswith(FP8_LINEAR_SUFFIXES) and out[name].dim() == 2:
            q, scale = quantize_rows_fp8(out[name])
            out[name] = q * scale[:, None]
    return out


# =============================================================================== vision
def vision_position_ids(grid_thw: torch.Tensor, merge: int) -> torch.Tensor:
    """(h, w) index of every patch, block-major over merge x merge groups
    (same scheme as transformers/vision_utils.py:81-127)."""
    out = []
    for t, h, w in grid_thw.tolist():
        hp = torch.arange(h).unsqueeze(1).expand(h, w)
        wp = torch.arange(w).unsqueeze(0).expand(h, w)
        shp = (h // merge, merge, w // merge, merge)
        hp = hp.reshape(shp).permute(0, 2, 1, 3).flatten()
        wp = wp.reshape(shp).permute(0, 2, 1, 3).flatten()
        out.append(torch.stack([hp, wp], dim=-1).repeat(t, 1))
    return torch.cat(out, dim=0)


def vision_rope_cos_sin(grid_thw: torch.Tensor, head_dim: int, merge: int, theta: float = 10000.0):
    """cos/sin [N, head_dim]: VisionRotaryEmbedding(head_dim//2) over (h, w), cat(emb, emb)."""
    dim = head_dim // 2
    inv_freq = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
    pos = vision_position_ids(grid_thw, merge).float()               # [N, 2]
    freqs = (pos.unsqueeze(-1) * inv_freq).flatten(1)                # [N, head_dim/2]
    emb = torch.cat((freqs, freqs), dim=-1)                          # [N, head_dim]
    return emb.cos(), emb.sin()


def _attention(q, k, v, scale: float, causal: bool, emu: bool, q_chunk: int = 2048):
    """q [H,Nq,D], k/v [H,Nk,D] fp32 -> [H,Nq,D].  Softmax in fp32; in emulate mode P is rounded
    to bf16 before the PV contraction (the row sum stays fp32), as the flash kernel does."""
    H, Nq, D = q.shape
    Nk = k.shape[1]
    out = torch.empty_like(q)
    for s in range(0, Nq, q_chunk):
        e = min(Nq, s + q_chunk)
        S = torch.matmul(q[:, s:e], k.transpose(1, 2)) * scale
        if causal:
            qi = torch.arange(s, e).unsqueeze(1) + (Nk - Nq)
            kj = torch.arange(Nk).unsqueeze(0)
            S = S.masked_fill(kj > qi, float("-inf"))
        m = S.max(dim=-1, keepdim=True).values
        P = torch.exp(S - m)
        l = P.sum(dim=-1, keepdim=True)
        out[:, s:e] = torch.matmul(_r(P, emu), v) / l
    return _r(out, emu)


def vision_block(sd: SD, p: str, x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, seqlens, n_heads: int, head_dim: int,
                 eps: float, emu: bool, a8: bool = False) -> torch.Tensor:
    """One pre-norm transformer block of the vision tower on packed patches x [N, E] (cos/sin [N, 1, head_dim]):
         x += proj(attention(rope(qkv(rmsnorm1(x)))))      bidirectional, one sequence per image
         x += fc2(silu(fc1(rmsnorm2(x))) * fc3(rmsnorm2(x)))
    Biases are used when the state dict has them.  PINNED (tests/test_oracle_pins.py) against the in-container
    transformers Qwen2_5_VLVisionBlock — the same composition with fc1/fc3/fc2 = gate/up/down — on shared weights."""
    E = n_heads * head_dim
    scale = 1.0 / math.sqrt(head_dim)
    h = rms_norm(x, _w(sd, p + "norm1.weight"), eps, emu)
    qb = sd.get(p + "attn.qkv.bias")
    qkv = _r(linear(h, _w(sd, p + "attn.qkv.weight"), None if qb is None else qb.float(), a8), emu)
    q, k, vv = qkv.view(-1, 3, n_heads, head_dim).unbind(1)    # [N,H,D]
    q = _r(q * cos + rotate_half(q) * sin, emu)
    k = _r(k * cos + rotate_half(k) * sin, emu)
    att = torch.empty_like(q)
    s0 = 0
    for n in seqlens:
        att[s0:s0 + n] = _attention(q[s0:s0 + n].transpose(0, 1), k[s0:s0 + n].transpose(0, 1),
                                    vv[s0:s0 + n].transpose(0, 1), scale, False, emu).transpose(0, 1)
        s0 += n
    pb = sd.get(p + "attn.proj.bias")
    x = _r(x + linear(att.reshape(-1, E), _w(sd, p + "attn.proj.weight"), None if pb is None else pb.float(), a8), emu)
    h = rms_norm(x, _w(sd, p + "norm2.weight"), eps, emu)
    b1, b2, b3 = (sd.get(p + f"mlp.fc{j}.bias") for j in (1, 2, 3))
    g = linear(h, _w(sd, p + "mlp.fc1.weight"), None if b1 is None else b1.float(), a8)
    u = linear(h, _w(sd, p + "mlp.fc3.weight"), None if b3 is None else b3.float(), a8)
    a = _r(F.silu(g) * u, emu)                     # fused epilogue: one rounding
    return _r(x + linear(a, _w(sd, p + "mlp.fc2.weight"), None if b2 is None else b2.float(), a8), emu)


def vision_tower(sd: SD, cfg, pixel_values: torch.Tensor, grid_thw: torch.Tensor,
                 emulate_bf16: bool = False, return_hidden: bool = False, a8: bool = False):
    """pixel_values [N, C*T*P*P] f32, grid_thw [n_img, 3] -> merged embeddings [N/merge^2, hidden].
    a8: per-token e4m3 activations into every block / merger linear (the patch embedding stays unquantised)."""
    v = cfg.vision
    emu = emulate_bf16
    E, Hh, D = v.embed_dim, v.num_attention_heads, v.head_dim
    pre = "vision_tower."
    x = _r(pixel_values.float(), emu)                  # the bf16 tower casts its input to bf16
    # patch embed: Conv2d(k=s=patch) on [N, C, P, P] (temporal slice 0) == GEMM over flattened patches
    x = x.view(-1, v.num_channels, v.temporal_patch_size, v.patch_size, v.patch_size)[:, :, 0].reshape(x.shape[0], -1)
    wpe = _w(sd, pre + "patch_embed.patchifier.proj.weight").reshape(E, -1)
    bpe = sd.get(pre + "patch_embed.patchifier.proj.bias")
    x = _r(linear(x, wpe, None if bpe is None else bpe.float()), emu)
    x = rms_norm(x, _w(sd, pre + "patch_embed.patchifier.norm.weight"), v.rms_norm_eps, emu)

    cos, sin = vision_rope_cos_sin(grid_thw, D, v.spatial_merge_size)
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)      # [N,1,D]
    seqlens = (grid_thw[:, 1] * grid_thw[:, 2]).repeat_interleave(grid_thw[:, 0]).tolist()
    scale = 1.0 / math.sqrt(D)
    hiddens = []
    for i in range(v.num_hidden_layers):
        x = vision_block(sd, f"{pre}blocks.{i}.", x, cos, sin, seqlens, Hh, D, v.rms_norm_eps, emu, a8)
        if return_hidden:
            hiddens.append(x.clone())
    if v.post_norm:
        x = rms_norm(x, _w(sd, pre + "post_trunk_norm.weight"), v.rms_norm_eps, emu)
    # PatchMerger (cf. modeling_qwen2_vl.py:277-290)
    x = layer_norm(x, _w(sd, pre + "merger.ln_q.weight"), _w(sd, pre + "merger.ln_q.bias"), v.merger_ln_eps, emu)
    x = x.view(-1, E * v.spatial_merge_size ** 2)
    x = _r(F.gelu(linear(x, _w(sd, pre + "merger.mlp.0.weight"), _w(sd, pre + "merger.mlp.0.bias"), a8)), emu)
    x = _r(linear(x, _w(sd, pre + "merger.mlp.2.weight"), _w(sd, pre + "merger.mlp.2.bias"), a8), emu)
    return (x, hiddens) if return_hidden else x


# =============================================================================== language model
def lm_rope_cos_sin(positions: torch.Tensor, head_dim: int, theta: float):
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    freqs = positions.float().unsqueeze(-1) * inv_freq
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


class KVCache:
    def __init__(self, n_layers: int):
        self.k: List[Optional[torch.Tensor]] = [None] * n_layers
        self.v: List[Optional[torch.Tensor]] = [None] * n_layers

    def append(self, i, k, v):
        self.k[i] = k if self.k[i] is None else torch.cat([self.k[i], k], dim=1)
        self.v[i] = v if self.v[i] is None else torch.cat([self.v[i], v], dim=1)
        return self.k[i], self.v[i]

    @property
    def length(self) -> int:
        return 0 if self.k[0] is None else self.k[0].shape[1]


def lm_forward(sd: SD, cfg, embeds: torch.Tensor, cache: KVCache, emulate_bf16: bool = False,
               last_only: bool = True, return_hidden: bool = False, a8: bool = False):
    """One sequence. embeds [T, hidden]; appends to cache; returns fp32 logits [1 or T, vocab].
    a8: per-token e4m3 activations into the seven layer linears (the fp8 engine's PREFILL; its decode step and the lm_head keep
    bf16 activations)."""
    emu = emulate_bf16
    T = embeds.shape[0]
    Hq, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    past = cache.length
    cos, sin = lm_rope_cos_sin(torch.arange(past, past + T), D, cfg.rope_theta)
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    x = embeds
    scale = 1.0 / math.sqrt(D)
    hiddens = []
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        h = rms_norm(x, _w(sd, p + "input_layernorm.weight"), cfg.rms_norm_eps, emu)
        bq, bk, bv = (sd.get(p + f"self_attn.{n}_proj.bias") for n in "qkv")
        q = _r(linear(h, _w(sd, p + "self_attn.q_proj.weight"), None if bq is None else bq.float(), a8), emu).view(T, Hq, D)
        k = _r(linear(h, _w(sd, p + "self_attn.k_proj.weight"), None if bk is None else bk.float(), a8), emu).view(T, Hkv, D)
        vv = _r(linear(h, _w(sd, p + "self_attn.v_proj.weight"), None if bv is None else bv.float(), a8), emu).view(T, Hkv, D)
        q = _r(q * cos + rotate_half(q) * sin, emu)
        k = _r(k * cos + rotate_half(k) * sin, emu)
        K, V = cache.append(i, k.transpose(0, 1), vv.transpose(0, 1))      # [Hkv, ctx, D]
        rep = Hq // Hkv
        att = _attention(q.transpose(0, 1), K.repeat_interleave(rep, dim=0), V.repeat_interleave(rep, dim=0),
                         scale, True, emu).transpose(0, 1).reshape(T, Hq * D)
        x = _r(x + linear(att, _w(sd, p + "self_attn.o_proj.weight"), None, a8), emu)
        h = rms_norm(x, _w(sd, p + "post_attention_layernorm.weight"), cfg.rms_norm_eps, emu)
        a = _r(F.silu(linear(h, _w(sd, p + "mlp.gate_proj.weight"), None, a8)) * linear(h, _w(sd, p + "mlp.up_proj.weight"), None, a8), emu)
        x = _r(x + linear(a, _w(sd, p + "mlp.down_proj.weight"), None, a8), emu)
        if return_hidden:
            hiddens.append(x.clone())
    if last_only:
        x = x[-1:]
    x = rms_norm(x, _w(sd, "model.norm.weight"), cfg.rms_norm_eps, emu)
    head = sd["lm_head.weight"] if "lm_head.weight" in sd else sd["model.embed_tokens.weight"]
    logits = linear(x, head.float())                                        # fp32 logits, never rounded
    return (logits, hiddens) if return_hidden else logits


def build_embeds(sd: SD, cfg, input_ids: torch.Tensor, vision_embeds: Optional[torch.Tensor]) -> torch.Tensor:
    """embed_tokens + masked_scatter of vision rows at image_token_id positions."""
    emb = sd["model.embed_tokens.weight"].float()[input_ids]
    if vision_embeds is not None:
        mask = input_ids == cfg.image_token_id
        assert int(mask.sum()) == vision_embeds.shape[0], "image token count != vision rows"
        emb = emb.clone()
        emb[mask] = vision_embeds
    return emb


@torch.no_grad()
def generate(sd: SD, cfg, input_ids: torch.Tensor, pixel_values: Optional[torch.Tensor],
             grid_thw: Optional[torch.Tensor], max_new_tokens: int, eos_ids: Tuple[int, ...] = (),
             emulate_bf16: bool = False, forced_tokens: Optional[List[int]] = None,
             return_logits: bool = False, vision_embeds: Optional[torch.Tensor] = None, fp8_act: bool = False):
    """Greedy decode of ONE sequence (reference parser.py:110 with do_sample=False).
    forced_tokens: teacher forcing (feed these instead of the argmax) so per-step logits can be
    compared with an engine that took a different branch at a near-tie.
    vision_embeds: merged vision rows computed elsewhere (skips the tower: LM-only comparisons at full context length).
    fp8_act: the fp8 engine's W8A8 phases — vision tower and PREFILL quantise the inputs of their linears per token; decode steps do
             not (weight-only there).  Use together with quantize_fp8_state_dict(sd).
    Returns (new_token_ids, [logits per step] if return_logits)."""
    vis = vision_embeds
    if vis is None and pixel_values is not None:
        vis = vision_tower(sd, cfg, pixel_values, grid_thw, emulate_bf16, a8=fp8_act)
    emb = build_embeds(sd, cfg, input_ids, vis)
    if emulate_bf16:
        emb = _r(emb, True)
    cache = KVCache(cfg.num_hidden_layers)
    logits = lm_forward(sd, cfg, emb, cache, emulate_bf16, a8=fp8_act)
    out, all_logits = [], []
    for step in range(max_new_tokens):
        if return_logits:
            all_logits.append(logits[0].clone())
        tok = int(torch.argmax(logits[0]))
        out.append(tok)
        if tok in eos_ids:
            break
        feed = tok if forced_tokens is None else forced_tokens[step]
        if step + 1 < max_new_tokens:
            e = sd["model.embed_tokens.weight"].float()[torch.tensor([feed])]
            logits = lm_forward(sd, cfg, e, cache, emulate_bf16)
    return (out, all_logits) if return_logits else out
