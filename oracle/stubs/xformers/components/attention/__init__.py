"""NystromAttention as the reference's NystromBlock calls it (layers/nystrom_attention.py:44-46,81): constructed with
(num_landmarks=128, num_heads, dropout), called with q, k, v of shape [b, n, h, d].  The arithmetic is the oracle's restatement of
the published algorithm (oracle/restate_v1.py nystrom_attention) applied per head over tokens -- PARITY UNPINNED, see there."""
import torch


class NystromAttention(torch.nn.Module):
    def __init__(self, num_landmarks=64, num_heads=1, dropout=0.0, **kwargs):
        super().__init__()
        self.num_landmarks = num_landmarks

    def forward(self, q, k, v, key_padding_mask=None, **kwargs):
        from oracle.restate_v1 import nystrom_attention
        b, n, h, d = q.shape

        def f(t):
            return t.permute(0, 2, 1, 3).reshape(b * h, t.shape[1], d)
        o = nystrom_attention(f(q), f(k), f(v), self.num_landmarks)
        return o.reshape(b, h, n, d).permute(0, 2, 1, 3)
