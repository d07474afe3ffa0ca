"""Host -> device input stage.

The reference's loop (clis/train_parallel.py:35-50, 189-199) converts every uint8 / float numpy array of the next
batch to **fp32 on the host**, copies it with ``non_blocking=True`` from pageable memory (which is not asynchronous)
and then normalises and permutes the image on the GPU in two more passes.  Here:

* the arrays travel in their own dtype (uint8 images: a quarter of the PCIe bytes) through **pinned, double-buffered**
  staging memory on a dedicated copy stream, so the transfer of batch i+1 overlaps the training step of batch i;
* ``mdm_input_stage`` turns the uint8 NHWC image into the normalised fp32 NCHW tensor in one kernel on that stream
  (``(u - 127) / 128``, the reference's formula);
* the consumer stream waits on the slot's event the first time the batch is touched -- no host synchronisation.

``load_batch(next_sample, device)`` keeps the reference's name, argument meaning and result keys (``image`` fp32 NHWC
is produced lazily, only if somebody asks for it); the batch additionally carries ``images`` (what the loop computes at
:194-195).  Precomputed text embeddings (``reader.py:107-112``, ``load_numpy``) are ordinary float arrays of the
sample and take the same path.
"""
import numpy as np
import torch

from . import ops


def _decode(codes):
    codes = np.asarray(codes).astype(np.uint8)
    return bytes(codes[codes != 0]).decode("latin-1")


class StagedBatch(dict):
    """dict of device tensors whose producer is the copy stream; the first access makes the current stream wait."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._event = None
        self._u8 = None

    def ready(self):
        ev, self._event = self._event, None
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            # the tensors were allocated on the copy stream: tell the caching allocator who else reads them
            for v in list(dict.values(self)) + [self._u8]:
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)
        return self

    # every way of getting a tensor out of the batch orders the consumer stream behind the copy stream first
    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return dict.items(self.ready())

    def values(self):
        return dict.values(self.ready())

    def keys(self):
        return dict.keys(self.ready())

    def __iter__(self):
        return dict.__iter__(self.ready())

    def pop(self, key, *default):
        return dict.pop(self.ready(), key, *default)

    def popitem(self):
        return dict.popitem(self.ready())

    def copy(self):
        return dict(self.ready())

    def __getitem__(self, key):
        self.ready()
        if key == "image" and not dict.__contains__(self, "image") and self._u8 is not None:
            dict.__setitem__(self, "image", self._u8.to(torch.float32))   # the reference's fp32 NHWC view, on demand
        return dict.__getitem__(self, key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or (key == "image" and self._u8 is not None)


class InputStager:
    """Pinned double-buffered H2D path with the uint8 -> normalised-image kernel on the copy stream."""

    def __init__(self, device, depth: int = 2):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ops._lib.MdmHipError("InputStager needs a GPU device (no CPU fallback)")
        self.stream = torch.cuda.Stream(self.device)
        self._slots = [dict(pinned={}, event=None) for _ in range(max(2, depth))]
        self._i = 0

    def _pinned(self, slot, key, arr):
        buf = slot["pinned"].get(key)
        if buf is None or buf.shape != arr.shape or buf.dtype != torch.from_numpy(arr[:0]).dtype:
            buf = torch.empty(arr.shape, dtype=torch.from_numpy(arr[:0]).dtype, pin_memory=True)
            slot["pinned"][key] = buf
        return buf

    def load_batch(self, next_sample):
        slot = self._slots[self._i % len(self._slots)]
        self._i += 1
        if slot["event"] is not None:
            slot["event"].synchronize()      # the copies that last read this slot's pinned memory (two batches ago)
        out = StagedBatch()
        with torch.cuda.stream(self.stream):
            for key, val in next_sample.items():
                if isinstance(val, np.ndarray) and val.dtype.kind in "uf":
                    arr = np.ascontiguousarray(val)
                    pin = self._pinned(slot, key, arr)
                    pin.copy_(torch.from_numpy(arr))
                    dev = pin.to(self.device, non_blocking=True)
                    if key == "image" and arr.dtype == np.uint8 and arr.ndim == 4 and arr.shape[-1] == 3:
                        out._u8 = dev
                        dict.__setitem__(out, "images", ops.input_stage(dev))
                    else:
                        dict.__setitem__(out, key, dev.to(torch.float32))
                else:
                    dict.__setitem__(out, key, val)
            if "watermark_score" in next_sample:
                # zero-padded character codes of a decimal string per sample (reader.py:199-202)
                ws = torch.tensor([float(_decode(w)) for w in next_sample["watermark_score"]])
                dict.__setitem__(out, "watermark_score", ws.to(self.device, non_blocking=True))
            if dict.__contains__(out, "state") and out._u8 is not None:
                # scale = image side / original size (:47-50); image is NHWC so size(2) is the width
                dict.__setitem__(out, "scale", float(out._u8.size(2)) / dict.__getitem__(out, "state")[:, 0])
            ev = torch.cuda.Event()
            ev.record(self.stream)
        slot["event"] = ev
        out._event = ev
        return out


_STAGERS = {}


def load_batch(next_sample, device):
    """Drop-in for clis/train_parallel.py:35-50 (same call, same keys); see the module docstring for what differs."""
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    st = _STAGERS.get(dev)
    if st is None:
        st = _STAGERS[dev] = InputStager(dev)
    return st.load_batch(next_sample)
