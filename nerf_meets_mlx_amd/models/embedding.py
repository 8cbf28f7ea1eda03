"""The frequency embedder NeRF really uses (`mlx_nerf/models/embedding.py:4-90`).

`ref_quirks=True` mirrors the k^2 frequency list; False gives 2^k.  On the
render path the embedding is fused into the MLP kernel; this standalone form exists for
API parity and tests."""
import torch

from .. import _native as N


class Embedder:
    def __init__(self, **kwargs) -> None:
        self.kwargs = kwargs
        self.create_embedding_func()

    def create_embedding_func(self):
        kw = self.kwargs
        self.in_dim = kw.get("input_dims", 3)
        self.n_freqs = kw["num_freqs"]
        if not kw["log_sampling"]:
            raise NotImplementedError                                         # embedding.py:51
        self.include_input = bool(kw["include_input"])                        # False only for 2-d inputs (embedding.py:79)
        self.freq_mode = 0 if kw.get("ref_quirks", True) else 1
        self.out_dim = self.in_dim * ((1 if self.include_input else 0) + 2 * self.n_freqs)

    def embed(self, inputs: torch.Tensor) -> torch.Tensor:
        x = N.f32(inputs)
        M = x.shape[0]
        out = torch.empty(M, self.in_dim * (1 + 2 * self.n_freqs), dtype=torch.float32, device=x.device)
        N.check(N.lib().nerf_encode_freq(N.ptr(x), M, self.in_dim, self.n_freqs, self.freq_mode, N.ptr(out), N.stream()))
        return out if self.include_input else out[:, self.in_dim:].contiguous()    # the kernel always writes the raw block first


def get_embedder(n_freqs: int, /, n_input_dims: int = 3, ref_quirks: bool = True):
    if n_freqs == -1:
        return (lambda x: x), 3
    eo = Embedder(include_input=False if 2 == n_input_dims else 3, input_dims=n_input_dims, max_freq_log2=n_freqs - 1,
                  num_freqs=n_freqs, log_sampling=True, ref_quirks=ref_quirks)
    fn = lambda x, eo=eo: eo.embed(x)
    fn.embedder = eo
    return fn, eo.out_dim


def embed(pos, embed_pos, dir, embed_dir):
    """Flatten positions, repeat directions to every sample, concat (`embedding.py:4-21`)."""
    pos_flat = pos.reshape(-1, pos.shape[-1])
    e_pos = embed_pos(pos_flat)
    if dir is None:
        return e_pos
    dirs = dir[:, None, :].expand(pos.shape[0], pos.shape[1], dir.shape[-1]).reshape(-1, dir.shape[-1])
    return torch.cat([e_pos, embed_dir(dirs)], dim=-1)
