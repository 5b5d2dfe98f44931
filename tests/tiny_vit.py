"""A tiny random ViT shared by tests/golden/gen_golden.py (which feeds it to the REFERENCE's FeatureExtractor) and the tests (which feed
it to hbird_mi.models.FeatureExtractor): none of the family APIs (no forward_features / get_intermediate_layers / config), only
`blocks[i].attn.qkv`, i.e. what the reference's generic fallback hooks (hbird/models.py:257-321).  The qkv module returns the 5-D
[B, N, 3, heads, Dh] tensor that the reference's unpacking expects (models.py:305); `flat=True` returns the usual [B, N, 3 * D] of an
nn.Linear instead (the reference cannot unpack that; the build reshapes it with attn.num_heads)."""
import torch
import torch.nn as nn


class _QKV(nn.Module):
    def __init__(self, d, heads, flat):
        super().__init__()
        self.lin, self.heads, self.flat = nn.Linear(d, 3 * d), heads, flat

    def forward(self, x):
        y = self.lin(x)
        if self.flat:
            return y
        B, N, _ = y.shape
        return y.view(B, N, 3, self.heads, -1)


class _Attn(nn.Module):
    def __init__(self, d, heads, flat):
        super().__init__()
        self.num_heads = heads
        self.qkv = _QKV(d, heads, flat)
        self.proj = nn.Linear(d, d)

    def forward(self, x):
        B, N, D = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, D // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = (q @ k.transpose(-2, -1)) * (D // self.num_heads) ** -0.5
        return self.proj((a.softmax(-1) @ v).transpose(1, 2).reshape(B, N, D))


class _Block(nn.Module):
    def __init__(self, d, heads, flat):
        super().__init__()
        self.n1, self.attn, self.n2 = nn.LayerNorm(d), _Attn(d, heads, flat), nn.LayerNorm(d)
        self.mlp = nn.Sequential(nn.Linear(d, 2 * d), nn.GELU(), nn.Linear(2 * d, d))

    def forward(self, x):
        x = x + self.attn(self.n1(x))
        return x + self.mlp(self.n2(x))


class TinyQKVViT(nn.Module):
    def __init__(self, d=16, heads=2, depth=2, ps=8, flat=False, seed=0):
        super().__init__()
        g = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.embed = nn.Conv2d(3, d, ps, ps)
        self.cls = nn.Parameter(torch.randn(1, 1, d) * 0.02)
        self.blocks = nn.ModuleList([_Block(d, heads, flat) for _ in range(depth)])
        torch.random.set_rng_state(g)

    def forward(self, imgs):
        x = self.embed(imgs).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls.expand(x.shape[0], -1, -1), x], dim=1)
        for b in self.blocks:
            x = b(x)
        return x
