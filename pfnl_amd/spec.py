"""Model geometry + checkpoint tensor layout of the PFNL forward path.

Mirrors the hard-coded configuration of the reference (`model/pfnl.py:21-37` for the
attributes, `model/pfnl.py:40-53` + `utils.py:23-26,66-67` for the variables).  Tensor names
follow the TF1 layers scoping rules: everything lives under the
variable scope ``nlvsr``; kernels are HWIO float32, biases are [Cout].

The 2x / T=5 geometry (BASELINE.json configs[4]) is *not expressible by the reference* (its tail
is hard-wired for 4x, `model/pfnl.py:52-53,76-78`); it is defined here as: convmerge2 12->3 and no
second depth_to_space.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple


@dataclass(frozen=True)
class PFNLGeometry:
    num_frames: int = 7      # model/pfnl.py:22
    scale: int = 4           # model/pfnl.py:23
    mf: int = 64             # model/pfnl.py:40
    num_block: int = 20      # model/pfnl.py:43
    in_ch: int = 3

    def __post_init__(self):
        if self.scale not in (2, 4):
            raise ValueError("scale must be 2 or 4")
        if self.num_frames < 1 or self.num_frames % 2 == 0:
            raise ValueError("num_frames must be odd")

    @property
    def stack_ch(self) -> int:          # channels of the frame stack, pfnl.py:55-56
        return self.num_frames * self.in_ch

    @property
    def nl_ch(self) -> int:             # channels after space_to_depth(2), pfnl.py:57-58
        return 4 * self.stack_ch

    @property
    def merge1_out(self) -> int:        # pfnl.py:52
        return 48

    @property
    def merge2_in(self) -> int:
        return self.merge1_out // 4

    @property
    def merge2_out(self) -> int:        # pfnl.py:53 (4x); build-defined for 2x
        return 12 if self.scale == 4 else 3

    def weight_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """Ordered (tf_name, shape) list — the checkpoint layout of SURVEY.md §8(a)-W."""
        mf, T = self.mf, self.num_frames
        out: List[Tuple[str, Tuple[int, ...]]] = []

        def add(name, kh, cin, cout):
            out.append((f"nlvsr/{name}/kernel", (kh, kh, cin, cout)))
            out.append((f"nlvsr/{name}/bias", (cout,)))

        add("conv0", 5, self.in_ch, mf)
        for i in range(self.num_block):
            add(f"conv1_{i}", 3, mf, mf)
        for i in range(self.num_block):
            add(f"conv10_{i}", 1, T * mf, mf)
        for i in range(self.num_block):
            add(f"conv2_{i}", 3, 2 * mf, mf)
        add("convmerge1", 3, T * mf, self.merge1_out)
        add("convmerge2", 3, self.merge2_in, self.merge2_out)
        add("nlblock_0/g/g", 1, self.nl_ch, self.nl_ch)
        add("nlblock_0/w/w", 1, self.nl_ch, self.nl_ch)
        return out

    def optional_weight_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """theta / phi 1x1 projections of the non-local block (reference utils.py:31-42: created for nltype 0 / 2 only;
        PFNL's nltype 1 checkpoint has none).  All four present = embedded-Gaussian block, none = the reference's."""
        C = self.nl_ch
        out: List[Tuple[str, Tuple[int, ...]]] = []
        for n in ("theta/theta", "phi/phi"):
            out.append((f"nlvsr/nlblock_0/{n}/kernel", (1, 1, C, C)))
            out.append((f"nlvsr/nlblock_0/{n}/bias", (C,)))
        return out

    def num_params(self) -> int:
        n = 0
        for _, s in self.weight_shapes():
            p = 1
            for d in s:
                p *= d
            n += p
        return n

    def flops_per_clip(self, H: int, W: int, shared_base: bool = False) -> float:
        """Algorithmic FLOPs (2*MACs) of the reference graph for one clip."""
        P = H * W
        N = (H // 2) * (W // 2)
        T, mf, C = self.num_frames, self.mf, self.nl_ch
        conv0 = T * 25 * self.in_ch * mf
        conv1 = T * 9 * mf * mf
        conv10 = T * mf * mf
        conv2 = T * 9 * 2 * mf * mf
        if shared_base:
            conv2 = (T + 1) * 9 * mf * mf
        block = conv1 + conv10 + conv2
        merge1 = 9 * T * mf * self.merge1_out
        merge2 = 4 * 9 * self.merge2_in * self.merge2_out
        macs = P * (conv0 + self.num_block * block + merge1 + merge2) + 2 * N * C * C + 2 * N * N * C
        return 2.0 * macs


DEFAULT_GEOMETRY = PFNLGeometry()


def check_weights(geom: PFNLGeometry, weights: Dict[str, "object"]) -> None:
    """Raise if a weight dict does not match the checkpoint layout."""
    for name, shape in geom.weight_shapes():
        if name not in weights:
            raise KeyError(f"missing tensor {name}")
        if tuple(weights[name].shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(weights[name].shape)}")
    opt = [(n, s) for n, s in geom.optional_weight_shapes() if n in weights]
    if opt and len(opt) != len(geom.optional_weight_shapes()):
        raise KeyError("nlblock_0 theta/phi: all four tensors or none")
    for name, shape in opt:
        if tuple(weights[name].shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(weights[name].shape)}")
