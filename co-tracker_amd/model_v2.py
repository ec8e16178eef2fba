"""CoTracker2 / CoTracker2.1 on the MI355X primitives.

Host-side mirror of ``CoTracker2`` (cotracker/models/core/cotracker/cotracker.py:29-384): same constructor kwargs,
attributes, ``forward`` signature / 3-tuple return, online-state methods and the same 321 ``state_dict`` keys
(``time_emb``, ``pos_emb``, ``fnet.*``, ``updateformer.*`` with 6 time + 6 space layers and a 130-wide ``flow_head``,
``norm.*``, ``track_feat_updater.0.*``, ``vis_predictor.0.*``), so reference checkpoints load unchanged.
"""
import torch
import torch.nn as nn

from .encoder import BasicEncoder
from .model import _Lin, _UpdateFormerParams, sincos_time_embed


def sincos_pos_embed_2d(dim: int, h: int, w: int) -> torch.Tensor:
    """get_2d_sincos_pos_embed (embeddings.py:11-55) -> [1, dim, h, w]: first half encodes the x (column) index,
    second half the y (row) index, each as [sin(pos*omega), cos(pos*omega)] with omega_k = 10000^(-k/(dim/4))."""
    def emb1d(d, pos):
        omega = torch.arange(d // 2, dtype=torch.double) / (d / 2.0)
        omega = 1.0 / 10000 ** omega
        out = torch.einsum("m,d->md", pos.reshape(-1).double(), omega)
        return torch.cat([torch.sin(out), torch.cos(out)], dim=1)
    gw, gh = torch.meshgrid(torch.arange(w, dtype=torch.float), torch.arange(h, dtype=torch.float), indexing="xy")
    emb = torch.cat([emb1d(dim // 2, gw), emb1d(dim // 2, gh)], dim=1).float()  # (h*w, dim): grid[0] = x index
    return emb.reshape(1, h, w, dim).permute(0, 3, 1, 2).contiguous()


class _Affine128(nn.Module):  # nn.GroupNorm(1, 128) parameters (cotracker.py:79)
    def __init__(self, dim=128):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class CoTracker2(nn.Module):
    """Constructor mirrors cotracker.py:30-84."""

    def __init__(self, window_len=8, stride=4, add_space_attn=True, num_virtual_tracks=64, model_resolution=(384, 512)):
        super().__init__()
        if num_virtual_tracks != 64 or not add_space_attn:
            raise NotImplementedError("HIP path is specialised to 64 virtual tracks with space attention on")
        self.window_len = window_len
        self.stride = stride
        self.hidden_dim = 256
        self.latent_dim = 128
        self.add_space_attn = add_space_attn
        self.num_virtual_tracks = num_virtual_tracks
        self.model_resolution = model_resolution
        self.input_dim = 456
        self.fnet = BasicEncoder(input_dim=3, output_dim=self.latent_dim, stride=stride)
        self.updateformer = _UpdateFormerParams(self.input_dim, 384, 6, num_virtual_tracks,
                                                flow_out=self.latent_dim + 2, vis_conf_head=False)
        self.register_buffer("time_emb", sincos_time_embed(self.input_dim, window_len))
        self.register_buffer("pos_emb", sincos_pos_embed_2d(self.input_dim, model_resolution[0] // stride,
                                                           model_resolution[1] // stride))
        self.norm = _Affine128(self.latent_dim)
        self.track_feat_updater = nn.Sequential(_Lin(self.latent_dim, self.latent_dim))  # + nn.GELU() (no parameters)
        self.vis_predictor = nn.Sequential(_Lin(self.latent_dim, 1))
