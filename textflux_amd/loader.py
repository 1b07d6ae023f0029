"""Checkpoint streaming: safetensors shards -> HBM (reference D/models/modeling_utils.py:468-1014, where
`from_pretrained` materialises every shard as a host state dict and copies tensor by tensor).

A 23.8 GB transformer is 780 tensors; loaded through `safetensors.safe_open(...).get_tensor` + pageable `tensor.copy_` each
tensor pays a page-cache -> heap copy, a pageable-H2D bounce inside the driver and a stream sync.  Here:

* the safetensors container is parsed directly (8-byte little-endian header length, JSON header with dtype / shape /
  data_offsets per tensor) and the data section is memory-mapped: no intermediate host tensors;
* tensors are visited in FILE ORDER (sequential reads) and cut into chunks of at most `chunk_bytes`; each chunk is copied
  page-cache -> one of `n_buffers` PINNED staging buffers (multi-threaded memcpy) and sent to its destination with an
  asynchronous H2D copy on a side stream; a buffer is reused when the event behind its last copy has fired, so the host
  memcpy of chunk i + 1 overlaps the DMA of chunk i;
* the destination is whatever device view the consumer names for (tensor, row range): for the DiT that is the row slice of
  the pre-fused matrices ([to_k; to_v; to_q], [k; v; q; mlp], the stacked modulation matrix), so the QKV / MLP
  concatenation IS the copy -- no torch.cat, no second pass;
* checkpoints that are not bf16 are sent as raw bytes to a device scratch buffer and converted there (`Tensor.to`, a dtype
  conversion, not arithmetic of the path).
On a CPU-only host (tests) the same code runs with plain copies."""
from __future__ import annotations

import json
import mmap
import os
import struct
import warnings
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch

_DT = {"BF16": torch.bfloat16, "F16": torch.float16, "F32": torch.float32, "F64": torch.float64, "I64": torch.int64,
       "I32": torch.int32, "I16": torch.int16, "I8": torch.int8, "U8": torch.uint8, "BOOL": torch.bool,
       "F8_E4M3": torch.float8_e4m3fn, "F8_E5M2": torch.float8_e5m2}


def read_header(path: str) -> Tuple[Dict[str, dict], int]:
    """(name -> {dtype, shape, data_offsets}, byte offset of the data section) of one .safetensors file."""
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        hdr = json.loads(f.read(n))
    hdr.pop("__metadata__", None)
    return hdr, 8 + n


def shard_files(root: str, stem: str) -> List[str]:
    """Files of a (possibly sharded) HF checkpoint: `<stem>.safetensors` or the `<stem>.safetensors.index.json` weight map."""
    index = os.path.join(root, stem + ".safetensors.index.json")
    if os.path.exists(index):
        with open(index) as f:
            return sorted(set(json.load(f)["weight_map"].values()))
    return [stem + ".safetensors"]


class ShardStreamer:
    def __init__(self, device, chunk_bytes: int = 256 << 20, n_buffers: int = 2):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.chunk = int(chunk_bytes)
        self.bytes_moved = 0
        if self.cuda:
            self.stream = torch.cuda.Stream(device=self.device)
            self.bufs = [torch.empty(self.chunk, dtype=torch.uint8).pin_memory() for _ in range(n_buffers)]
            self.events: List[Optional[torch.cuda.Event]] = [None] * n_buffers
            self.scratch: Optional[torch.Tensor] = None
        self._turn = 0

    def _send(self, src_u8: torch.Tensor, dst: torch.Tensor, src_dtype: torch.dtype) -> None:
        """src_u8: host uint8 view of the tensor's bytes (any length); dst: contiguous device tensor of the same element count."""
        n = src_u8.numel()
        self.bytes_moved += n
        if not self.cuda:
            dst.copy_(src_u8.view(src_dtype).view(dst.shape) if src_dtype == dst.dtype else src_u8.view(src_dtype).view(dst.shape).to(dst.dtype))
            return
        same = src_dtype == dst.dtype
        es = torch.empty(0, dtype=src_dtype).element_size()
        dst_flat = dst.view(-1)
        step = self.chunk // es * es
        for o in range(0, n, step):
            m = min(step, n - o)
            b = self._turn % len(self.bufs)
            self._turn += 1
            if self.events[b] is not None:
                self.events[b].synchronize()                       # the DMA that last read this staging buffer has finished
            self.bufs[b][:m].copy_(src_u8[o:o + m])               # page cache -> pinned (multi-threaded memcpy)
            with torch.cuda.stream(self.stream):
                if same:
                    dst_flat[o // es:(o + m) // es].view(torch.uint8).copy_(self.bufs[b][:m], non_blocking=True)
                else:                                               # raw bytes to device scratch, converted there
                    if self.scratch is None or self.scratch.numel() < step:
                        self.scratch = torch.empty(step, dtype=torch.uint8, device=self.device)
                    self.scratch[:m].copy_(self.bufs[b][:m], non_blocking=True)
                    dst_flat[o // es:(o + m) // es].copy_(self.scratch[:m].view(src_dtype))
                ev = torch.cuda.Event()
                ev.record(self.stream)
                self.events[b] = ev

    def stream_file(self, path: str, route: Callable[[str, Tuple[int, ...], torch.dtype], Optional[torch.Tensor]]) -> List[str]:
        """Every tensor of `path`, in file order: route(name, shape, dtype) returns the contiguous device (view) tensor that
        receives it -- same element count, any dtype -- or None to skip.  Returns the names seen."""
        hdr, base = read_header(path)
        names = sorted(hdr, key=lambda k: hdr[k]["data_offsets"][0])
        with open(path, "rb") as f:
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        try:
            if hasattr(mm, "madvise"):
                mm.madvise(mmap.MADV_SEQUENTIAL)
            arr = np.frombuffer(mm, dtype=np.uint8)
            for k in names:
                e = hdr[k]
                dt = _DT.get(e["dtype"])
                if dt is None:
                    raise RuntimeError(f"{path}: tensor {k} has unsupported dtype {e['dtype']}")
                dst = route(k, tuple(e["shape"]), dt)
                if dst is None:
                    continue
                lo, hi = e["data_offsets"]
                if int(np.prod(e["shape"], dtype=np.int64)) != dst.numel() or not dst.is_contiguous():
                    raise RuntimeError(f"{path}: destination of {k} must be contiguous with {e['shape']} elements")
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")               # read-only mapping: from_numpy warns, nothing writes to it
                    src = torch.from_numpy(arr[base + lo: base + hi])
                self._send(src, dst, dt)
            self.finish()        # the mapping must outlive the last host-side read (the pinned copies are synchronous, the DMAs not)
            del arr
        finally:
            try:
                mm.close()
            except BufferError:
                pass
        return names

    def finish(self) -> None:
        if self.cuda:
            self.stream.synchronize()
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
