"""TEST INFRASTRUCTURE ONLY.  A literal restatement of the parts of xformers that the reference's NystromBlock executes
(layers/nystrom_attention.py:8,44-46,81: `NystromAttention(num_landmarks=128, num_heads=num_heads, dropout=dropout)` called as
`self.attention_fn(q, k, v, key_padding_mask=attn_bias)` with q, k, v of shape [b, n, h, d]).

xformers is a third-party dependency of the reference (requirements.txt:24 `xformers>=0.0.26`), un-vendored and absent from this image (no
network).  What follows restates, statement by statement, the published source of xformers v0.0.26:

    xformers/components/attention/nystrom.py   AvgPool.forward, NystromAttention.__init__ / .forward
    xformers/components/attention/core.py      scaled_query_key_softmax, scaled_dot_product_attention, _matmul_with_mask, _softmax, bmm
    xformers/components/attention/utils.py     iterative_pinv

restricted to dense tensors without masks (the reference passes key_padding_mask = attn_bias = None and builds the module with causal=False,
no skip connection, dropout 0 at inference), every tensor operation kept in xformers' own order and with xformers' own dimension indices --
`size(-2)`, `shape[1]`, `shape[2]`, `transpose(-2, -1)` -- because WHICH axis each of them picks up for a 4-D input is the whole question.

What the reference's layout does to this code (the load-bearing lines are marked [*] below):

    q, k, v are [b, n, h, d].  NystromAttention.forward reads `seq_len = k.size(-2)` [*] = h, the HEAD count (4 in layers_8, 2 in layers_4:
    unidepthv1/decoder.py NystromBlock(num_heads = num_heads // 2 | // 4)), so `self.num_landmarks >= seq_len` [*] (128 >= h) is TRUE and the
    module takes its small-sequence branch: plain `scaled_dot_product_attention(q, k, v, att_mask=None)`, whose matmuls act on the LAST TWO
    axes: att = softmax((q / sqrt(d)) @ k.transpose(-2, -1)) is [b, n, h, h] and att @ v is [b, n, h, d].  I.e. the deployed layer is an
    exact softmax attention of every token's h head-vectors among THEMSELVES -- no landmark, no pseudo-inverse, and no mixing between tokens
    at all.  The Nystrom branch (landmark pooling, three kernels, iterative_pinv) is restated too, and is never reached with this layout.

The package itself cannot be executed here, so this restatement is the pin; the two lines it hinges on are the `seq_len = k.size(-2)` read and
the `>=` comparison, both quoted verbatim from nystrom.py.  oracle/restate_v1.py follows THIS module (the reference), not the paper."""
import math
from typing import Optional

import torch
import torch.nn as nn


# ---------------------------------------------------------------- xformers/components/attention/core.py
def _matmul_with_mask(a: torch.Tensor, b: torch.Tensor, mask: Optional[torch.Tensor]) -> torch.Tensor:
    if mask is None:
        return a @ b
    att = a @ b
    if mask.dtype == torch.bool:
        att = att.masked_fill(~mask, float("-inf"))
    else:
        att = att + mask
    return att


def _softmax(a: torch.Tensor, causal: bool = False) -> torch.Tensor:
    return torch.softmax(a, dim=a.ndim - 1)


def bmm(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return a @ b


def scaled_query_key_softmax(q: torch.Tensor, k: torch.Tensor, att_mask: Optional[torch.Tensor]) -> torch.Tensor:
    # Self-attend: (N, S, hs) x (N, hs, S) -> (N, S, S)
    q = q / math.sqrt(k.size(-1))
    att = _matmul_with_mask(q, k.transpose(-2, -1), att_mask)
    att = _softmax(att, causal=False)
    return att


def scaled_dot_product_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, att_mask: Optional[torch.Tensor], dropout=None) -> torch.Tensor:
    att = scaled_query_key_softmax(q, k, att_mask=att_mask)
    att = dropout(att) if dropout is not None else att
    # y = att @ v  # (N, S, S) x (N, S, hs) -> (N, S, hs)
    y = bmm(att, v)
    return y


# ---------------------------------------------------------------- xformers/components/attention/utils.py
def iterative_pinv(softmax_mat: torch.Tensor, n_iter=6, pinverse_original_init=False):
    """Moore-Penrose inverse by the iteration of Razavi et al. 2014."""
    i = torch.eye(softmax_mat.size(-1), device=softmax_mat.device, dtype=softmax_mat.dtype)
    k = softmax_mat
    # The entries of K are positive and ||K||_{\\infty} = 1 due to softmax
    if pinverse_original_init:
        # This original implementation is more conservative to compute coefficient of Z_0.
        v = 1 / torch.max(torch.sum(k, dim=-2)) * k.transpose(-1, -2)
    else:
        # This is the exact coefficient computation, 1 / ||K||_1, of initialization of Z_0, leading to faster convergence.
        v = 1 / torch.max(torch.sum(k, dim=-2), dim=-1).values[:, None, None] * k.transpose(-1, -2)
    for _ in range(n_iter):
        kv = torch.matmul(k, v)
        v = torch.matmul(0.25 * v, 13 * i - torch.matmul(kv, 15 * i - torch.matmul(kv, 7 * i - kv)))
    return v


# ---------------------------------------------------------------- xformers/components/attention/nystrom.py
class AvgPool(nn.Module):
    def __init__(self, n: int):
        super().__init__()
        self.n = n

    def forward(self, x: torch.Tensor):
        # Average independently for every segment in the sequence dimension
        seq_len = x.shape[1]
        head_dim = x.shape[2]
        segments = seq_len // self.n
        assert segments > 0, "num_landmarks should be smaller than the sequence length"

        # Dimensions are a match
        if seq_len % self.n == 0:
            return x.reshape(-1, self.n, segments, head_dim).mean(dim=-2)

        # Handle the last segment boundary being off
        n_round = self.n - seq_len % self.n
        x_avg_round = x[:, : n_round * segments, :].reshape(-1, n_round, segments, head_dim).mean(dim=-2)
        x_avg_off = x[:, n_round * segments:, :].reshape(-1, self.n - n_round, segments + 1, head_dim).mean(dim=-2)
        return torch.cat((x_avg_round, x_avg_off), dim=-2)


class NystromAttention(nn.Module):
    def __init__(self, dropout: float, num_heads: int, num_landmarks: int = 64, landmark_pooling: Optional[nn.Module] = None,
                 causal: bool = False, use_razavi_pinverse: bool = True, pinverse_original_init: bool = False, inv_iterations: int = 6,
                 v_skip_connection: Optional[nn.Module] = None, conv_kernel_size: Optional[int] = None, *args, **kwargs):
        super().__init__()
        self.requires_separate_masks = True
        self.num_landmarks = num_landmarks
        self.num_heads = num_heads
        self.use_razavi_pinverse = use_razavi_pinverse
        self.pinverse_original_init = pinverse_original_init
        self.inv_iterations = inv_iterations
        self.attn_drop = nn.Dropout(dropout)
        self.skip_connection = v_skip_connection
        self.causal = causal
        assert conv_kernel_size is None and v_skip_connection is None and not causal, "restated for the reference's construction only"
        if landmark_pooling is not None:
            self.landmark_pooling = landmark_pooling
        else:
            self.landmark_pooling = AvgPool(n=self.num_landmarks)
        self.supports_attention_mask = False
        self.supports_key_padding_mask = True
        self.last_branch = None             # not in xformers: which branch the last call took ("full" | "nystrom"), read by tests

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, key_padding_mask: Optional[torch.Tensor] = None, *args, **kwargs):
        assert key_padding_mask is None, "restated for the reference's call only (attn_bias = None)"
        batched_dim = k.size(0)                                               # noqa: F841  (used by the mask paths only)
        seq_len = k.size(-2)                                                   # [*]  4-D [b, n, h, d] input: this is h
        if self.num_landmarks >= seq_len:                                      # [*]  128 >= h: always, for the reference
            mask: Optional[torch.Tensor] = None
            x = scaled_dot_product_attention(q=q, k=k, v=v, att_mask=mask)
            self.last_branch = "full"
        else:
            q_landmarks = self.landmark_pooling(q)
            k_landmarks = self.landmark_pooling(k)
            mask_3: Optional[torch.Tensor] = None
            kernel_1 = scaled_query_key_softmax(q=q, k=k_landmarks, att_mask=None)
            kernel_2 = scaled_query_key_softmax(q=q_landmarks, k=k_landmarks, att_mask=None)
            kernel_3 = scaled_dot_product_attention(q=q_landmarks, k=k, v=v, att_mask=mask_3)
            kernel_2_inv = (iterative_pinv(kernel_2, self.inv_iterations, self.pinverse_original_init)
                            if self.use_razavi_pinverse else torch.linalg.pinv(kernel_2))
            x = torch.matmul(torch.matmul(kernel_1, kernel_2_inv), kernel_3)
            self.last_branch = "nystrom"
        x = self.attn_drop(x)
        return x
