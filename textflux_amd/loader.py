"""Checkpoint streaming: safetensors shards -> HBM (reference D/models/modeling_utils.py:468-1014, where
`from_pretrained` materialises every shard as a host state dict and copies tensor by tensor).

A 23.8 GB transformer is 780 tensors; loaded through `safetensors.safe_open(...).get_tensor` + pageable `tensor.copy_` each
tensor pays a page-cache -> heap copy, a pageable-H2D bounce inside the driver and a stream sync.  Here:

* the safetensors container is parsed directly (8-byte little-endian header length, JSON header with dtype / shape /
  data_offsets per tensor): no intermediate host tensors;
* the data section is read FRONT TO BACK in chunks of up to `chunk_bytes` (whatever tensors or pieces of tensors a chunk
  covers) straight into one of `n_buffers` PINNED staging buffers by `n_threads` positional reads in parallel (a single
  thread copying out of the page cache -- or faulting in a memory map, the first version of this loader -- delivers 3-4 GB/s),
  then every covered piece goes to its destination with an asynchronous H2D copy on a side stream; a buffer is reused when
  the event behind its last copy has fired, so the reads of chunk i + 1 overlap the DMA of chunk i;
* the destination is whatever device view the consumer names for (tensor, row range): for the DiT that is the row slice of
  the pre-fused matrices ([to_k; to_v; to_q], [k; v; q; mlp], the stacked modulation matrix), so the QKV / MLP
  concatenation IS the copy -- no torch.cat, no second pass;
* checkpoints that are not bf16 are sent as raw bytes to a device scratch buffer and converted there (`Tensor.to`, a dtype
  conversion, not arithmetic of the path).
On a CPU-only host (tests) the same code runs with plain copies."""
from __future__ import annotations

import json
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
    def __init__(self, device, chunk_bytes: Optional[int] = None, n_buffers: Optional[int] = None, n_threads: Optional[int] = None):
        # defaults from tools/loader_bench.py --sweep on an MI355X box (page-cache resident shards); TFX_LOADER = "threads,chunk MiB,buffers"
        env = [int(v) for v in os.environ.get("TFX_LOADER", "").split(",") if v.strip()]
        n_threads = n_threads or (env[0] if len(env) > 0 else 16)
        chunk_bytes = chunk_bytes or ((env[1] << 20) if len(env) > 1 else (128 << 20))
        n_buffers = n_buffers or (env[2] if len(env) > 2 else 3)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.chunk = int(chunk_bytes)
        self.bytes_moved = 0
        self.n_threads = max(1, int(n_threads))
        self._pool = None
        if self.cuda:
            self.stream = torch.cuda.Stream(device=self.device)
            self.bufs = [torch.empty(self.chunk, dtype=torch.uint8).pin_memory() for _ in range(n_buffers)]
        else:
            self.bufs = [torch.empty(min(self.chunk, 64 << 20), dtype=torch.uint8)]
            self.chunk = self.bufs[0].numel()
        self.events: List[Optional["torch.cuda.Event"]] = [None] * len(self.bufs)
        self.scratch: Optional[torch.Tensor] = None
        self._turn = 0

    # ---- file range -> staging buffer: positional reads from several threads (the GIL is released inside os.preadv; one thread
    # copies ~3-4 GB/s out of the page cache, the copy engine takes > 40 GB/s)
    def _fill(self, fd: int, file_off: int, n: int, buf: torch.Tensor) -> None:
        mv = memoryview(buf.numpy())
        piece = max(4 << 20, (n + self.n_threads - 1) // self.n_threads)
        piece = (piece + 4095) // 4096 * 4096

        def rd(o):
            m = min(piece, n - o)
            got = 0
            while got < m:
                k = os.preadv(fd, [mv[o + got:o + m]], file_off + o + got)
                if k <= 0:
                    raise IOError("short read from checkpoint shard")
                got += k

        offs = list(range(0, n, piece))
        if len(offs) == 1 or self.n_threads == 1:
            for o in offs:
                rd(o)
            return
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(self.n_threads)
        list(self._pool.map(rd, offs))

    def stream_file(self, path: str, route: Callable[[str, Tuple[int, ...], torch.dtype], Optional[torch.Tensor]]) -> List[str]:
        """Every tensor of `path`: route(name, shape, dtype) returns the contiguous device (view) tensor that receives it -- same
        element count, any dtype -- or None to skip.  The data section is read front to back in chunks of up to `chunk_bytes`
        (whatever tensors or tensor pieces a chunk covers), each chunk into the next pinned buffer, then one asynchronous
        copy per covered piece.  Returns the names seen."""
        hdr, base = read_header(path)
        names = sorted(hdr, key=lambda k: hdr[k]["data_offsets"][0])
        todo = []                                                   # (lo, hi, dst, src dtype)
        for k in names:
            e = hdr[k]
            dt = _DT.get(e["dtype"])
            if dt is None:
                raise RuntimeError(f"{path}: tensor {k} has unsupported dtype {e['dtype']}")
            dst = route(k, tuple(e["shape"]), dt)
            if dst is None:
                continue
            lo, hi = e["data_offsets"]
            if int(np.prod(e["shape"], dtype=np.int64)) != dst.numel() or not dst.is_contiguous() or hi - lo != dst.numel() * dt.itemsize:
                raise RuntimeError(f"{path}: destination of {k} must be contiguous with {e['shape']} elements")
            if hi > lo:
                todo.append((lo, hi, dst, dt))
        if self.cuda:
            # the destinations were allocated on the CURRENT stream: if the caching allocator handed out blocks that kernels queued
            # there are still using (a transformer swapped right after dropping the old one), the side-stream DMAs must queue behind them
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
        fd = os.open(path, os.O_RDONLY)
        try:
            i = 0
            while i < len(todo):
                c0 = todo[i][0]
                # extend the chunk over as many whole tensors as fit; a tensor larger than what is left is cut on an 8-byte boundary
                c1, j = c0, i
                while j < len(todo) and todo[j][0] - c0 < self.chunk:
                    lo, hi = todo[j][0], todo[j][1]
                    if hi - c0 <= self.chunk:
                        c1, j = hi, j + 1
                    else:
                        cut = lo + (c0 + self.chunk - lo) // 8 * 8 if lo >= c0 else c0 + self.chunk // 8 * 8
                        c1 = max(c1, cut)
                        break
                if c1 == c0:        # first tensor alone exceeds the chunk and starts before it: plain cut
                    c1 = c0 + self.chunk // 8 * 8
                b = self._turn % len(self.bufs)
                self._turn += 1
                if self.events[b] is not None:
                    self.events[b].synchronize()                    # the DMAs that last read this staging buffer have finished
                    self.events[b] = None
                self._fill(fd, base + c0, c1 - c0, self.bufs[b])
                self.bytes_moved += c1 - c0
                ctx = torch.cuda.stream(self.stream) if self.cuda else _Null()
                with ctx:
                    k = i
                    while k < len(todo) and todo[k][0] < c1:
                        lo, hi, dst, dt = todo[k]
                        s0, s1 = max(lo, c0), min(hi, c1)
                        self._put(self.bufs[b][s0 - c0:s1 - c0], dst, dt, s0 - lo)
                        if hi <= c1:
                            k += 1
                        else:
                            break
                    if self.cuda:
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                        self.events[b] = ev
                # next chunk starts where this one ended (inside tensor k if it was cut)
                if k < len(todo) and todo[k][0] < c1 < todo[k][1]:
                    lo, hi, dst, dt = todo[k]
                    todo[k] = (c1, hi, dst.view(-1)[(c1 - lo) // dt.itemsize:], dt) if (c1 - lo) % dt.itemsize == 0 else None
                    if todo[k] is None:
                        raise RuntimeError("internal: chunk boundary inside an element")
                i = k
            self.finish()
        finally:
            os.close(fd)
        return names

    def _put(self, src_u8: torch.Tensor, dst: torch.Tensor, src_dtype: torch.dtype, byte_off: int) -> None:
        """Staged bytes -> elements [byte_off / itemsize, ...) of the flat destination."""
        es = src_dtype.itemsize
        n = src_u8.numel()
        flat = dst.view(-1)[byte_off // es:(byte_off + n) // es]
        if not self.cuda:
            flat.copy_(src_u8.view(src_dtype) if src_dtype == dst.dtype else src_u8.view(src_dtype).to(dst.dtype))
        elif src_dtype == dst.dtype:
            flat.view(torch.uint8).copy_(src_u8, non_blocking=True)
        else:                                                       # raw bytes to device scratch, converted there
            if self.scratch is None or self.scratch.numel() < n:
                self.scratch = torch.empty(max(n, 1 << 20), dtype=torch.uint8, device=self.device)
            self.scratch[:n].copy_(src_u8, non_blocking=True)
            flat.copy_(self.scratch[:n].view(src_dtype))

    def finish(self) -> None:
        if self.cuda:
            self.stream.synchronize()
            torch.cuda.current_stream(self.device).wait_stream(self.stream)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
