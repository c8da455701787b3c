"""Batched per-step weight preparation.

Every convolution needs derived copies of its filter once per step: the three bf16 planes of the fp32 filter for the
pre-split x6 kernel (fprop), the transposed filter [Cin][kh][kw][Cout] for dgrad, and that one's planes.  Done per layer
these are two or three tiny launches each (DenseNet-121: 359 launches, 1.2 ms of a 25 ms step at batch 64; ResNet-50 at
batch 1024: 0.9 ms per step).  Here the layers REGISTER what they need the first time they ask for it (and serve
themselves through the per-layer kernels for that one step); from the next forward pass on, `Network.forward` runs ONE
kernel over a device-side segment table (csrc/hip/conv_x6p.hip: weight_prep_kernel) before the first layer and the
layers pick up views of the prepared buffers.

Validity: the prepared copies are made at the start of every forward pass (training or evaluation) from the arena the
kernels read, so they are exactly what the per-layer path would have produced - a layer's weights change only in the
optimizer step of its own bucket, which is queued after that layer's dgrad.  A Context that did not go through
`Network.forward` (layer-level tests) has no `wprep` and takes the per-layer path.
"""
from __future__ import annotations

import os

import torch

from .. import ops

ENABLED = os.environ.get("DPA_WEIGHT_PREP", "1") != "0"


class WeightPrep:
    def __init__(self, store):
        self.store = store
        self.req = {}            # key -> dict(spec, K, RS, C, transposed, planes, dst, pl)
        self.order = []
        self.dirty = False
        self.table = None
        self._old_tables = []
        self.total_chunks = 0
        self.ready = set()       # keys covered by the table of the most recent run
        self.src_ptr = None      # data_ptr of the compute arena the table was built for

    # ------------------------------------------------------------------ layer side
    def get(self, ctx, key, spec, krsc, transposed, planes):
        """Buffers for `key` if this forward pass prepared them, else None (after registering the request).
        krsc: the (K, RS, C) view of the filter the caller uses (a Linear-as-GEMM passes (K, 1, R*S*C))."""
        if getattr(ctx, "wprep", None) is not self:
            return None
        r = self.req.get(key)
        same = r is not None and (r["K"], r["RS"], r["C"]) == tuple(krsc)
        if same and (r["planes"] or not planes):        # planes are sticky: a pass that does not need them (a smaller
            if key in self.ready:                       # last batch) must not trigger a rebuild of the table
                return r
            return None
        if r is None:
            self.order.append(key)
        keep = r if same else None                      # upgrade (planes wanted now): keep the transposed buffer
        self.req[key] = dict(spec=spec, K=krsc[0], RS=krsc[1], C=krsc[2], transposed=transposed, planes=planes,
                             dst=keep["dst"] if keep else None, pl=None)
        self.ready.discard(key)
        self.dirty = True
        return None

    # ------------------------------------------------------------------ step side
    def run(self, ctx):
        """Called by Network.forward before the first layer."""
        ctx.wprep = None
        if not (ENABLED and ctx.cuda):
            return
        st = self.store
        arena = st.p16 if st.p16 is not None else st.p
        if self.dirty or (self.table is not None and self.src_ptr != arena.data_ptr()):
            if torch.cuda.is_current_stream_capturing():
                return                      # no allocation / host-to-device copy inside a capture: per-layer path this time
            self._rebuild(arena)
        ctx.wprep = self
        if self.table is not None:
            ops.weight_prep(self.table, self.table.shape[0], self.total_chunks, arena.dtype)

    def _rebuild(self, arena):
        dev, dt = arena.device, arena.dtype
        esz = arena.element_size()
        rows, chunk = [], 0
        for key in self.order:
            r = self.req[key]
            n = r["K"] * r["RS"] * r["C"]
            assert n % 8 == 0 and r["K"] % 8 == 0 and n == r["spec"].numel
            if r["transposed"] and r["dst"] is None:
                r["dst"] = torch.empty((r["C"], r["RS"], r["K"]), device=dev, dtype=dt)
            if r["planes"] and r["pl"] is None:
                r["pl"] = torch.empty(3 * n, device=dev, dtype=torch.bfloat16)
            src = arena.data_ptr() + r["spec"].offset * esz
            mode = 0
            if r["transposed"]:       # 2: through an LDS tile (fp32, 64 x 32 tiles), 1: strided gather
                mode = 2 if (esz == 4 and r["K"] % 64 == 0 and r["C"] % 32 == 0) else 1
            rows.append([chunk, n, src, r["dst"].data_ptr() if r["dst"] is not None else 0,
                         r["pl"].data_ptr() if r["pl"] is not None else 0,
                         r["K"] | (r["RS"] << 32), r["C"] | (mode << 32)])
            chunk += (n // 8 + 255) // 256 * 256          # a block of 256 chunks never straddles two segments
        if self.table is not None:
            self._old_tables.append(self.table)        # a captured hipGraph may still point at it
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev) if rows else None
        self.total_chunks = chunk
        self.ready = set(self.order)
        self.dirty = False
        self.src_ptr = arena.data_ptr()
