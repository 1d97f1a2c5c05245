"""Device-resident sliding-window data set and loader: the step BEFORE the hot path, on the GPU.

Mirrors ``ClimateDataset`` (reference main_final.py:46-154) and the ``DataLoader(shuffle=True)`` the DataModule builds
around it (main_final.py:483-494): sample ``idx`` is ``(inputs[idx-T+1 .. idx], outputs[idx])`` with all-zero frames
(in normalised space) where the window reaches before the start of the data (main_final.py:76,127-131); ``len`` is the
number of time steps.  The reference materialises every sample with a python loop + ``torch.stack`` on the host and
copies each collated batch (13 MB at BASELINE config 2) over PCIe; here the normalised arrays live in HBM (288 GB: the
whole 367 MB data set fits thousands of times) and one launch (``cm_build_windows``) gathers a batch straight into
the trainer's input buffers.  Index ORDER is torch's own: ``RandomSampler`` / ``BatchSampler`` are used as they are,
so an epoch visits the same batches as the reference's DataLoader under the same generator.
"""
from typing import Iterator, Optional, Tuple

import torch
from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler

from ._lib import check, lib


class DeviceWindowDataset:
    """inputs [N, C, H, W], outputs [N, C_out, H, W] (normalised, fp32) -> windows of ``seq_len`` frames."""

    def __init__(self, inputs: torch.Tensor, outputs: torch.Tensor, seq_len: int, device=None):
        if inputs.dim() != 4 or outputs.dim() != 4 or inputs.shape[0] != outputs.shape[0]:
            raise ValueError("expected inputs [N,C,H,W] and outputs [N,C_out,H,W] with the same N")
        if seq_len < 1:
            raise ValueError("seq_len must be >= 1")
        device = torch.device(device if device is not None else "cuda")
        if device.type != "cuda":
            raise RuntimeError("DeviceWindowDataset keeps the data set in HBM: it needs a GPU (no CPU fallback)")
        self.input_tensors = inputs.to(device=device, dtype=torch.float32).contiguous()      # names as the reference
        self.output_tensors = outputs.to(device=device, dtype=torch.float32).contiguous()
        self.seq_len = int(seq_len)
        self.total_timesteps = int(inputs.shape[0])
        self.size = self.total_timesteps

    def __len__(self) -> int:
        return self.size

    def batch_into(self, idx: torch.Tensor, x: torch.Tensor, y: torch.Tensor) -> None:
        """Gather the samples ``idx`` (int64, on the device) into x [B,T,C,H,W] / y [B,C_out,H,W]."""
        b = idx.numel()
        if tuple(x.shape) != (b, self.seq_len) + tuple(self.input_tensors.shape[1:]) or \
                tuple(y.shape) != (b,) + tuple(self.output_tensors.shape[1:]):
            raise RuntimeError("batch buffers do not match (B, seq_len, C, H, W) / (B, C_out, H, W)")
        if not (x.is_contiguous() and y.is_contiguous() and idx.is_cuda and idx.dtype == torch.int64):
            raise RuntimeError("batch_into needs contiguous device buffers and int64 device indices")
        check(lib.cm_build_windows(self.input_tensors.data_ptr(), self.output_tensors.data_ptr(), idx.data_ptr(),
                                   x.data_ptr(), y.data_ptr(), b, self.seq_len, self.input_tensors[0].numel(),
                                   self.output_tensors[0].numel(), self.total_timesteps,
                                   torch.cuda.current_stream().cuda_stream), "build_windows")

    def batch(self, idx) -> Tuple[torch.Tensor, torch.Tensor]:
        idx = torch.as_tensor(idx, dtype=torch.int64).to(self.input_tensors.device)
        x = torch.empty((idx.numel(), self.seq_len) + tuple(self.input_tensors.shape[1:]),
                        device=idx.device, dtype=torch.float32)
        y = torch.empty((idx.numel(),) + tuple(self.output_tensors.shape[1:]), device=idx.device, dtype=torch.float32)
        self.batch_into(idx, x, y)
        return x, y

    def __getitem__(self, i: int):
        x, y = self.batch([int(i)])
        return x[0], y[0]


class DeviceLoader:
    """``DataLoader(dataset, batch_size, shuffle, drop_last=False)`` for a DeviceWindowDataset: torch's samplers decide
    the order; batches are gathered on the device, optionally straight into a trainer's static input buffers."""

    def __init__(self, dataset: DeviceWindowDataset, batch_size: int, shuffle: bool = False, drop_last: bool = False,
                 generator: Optional[torch.Generator] = None, trainer=None):
        self.dataset, self.batch_size, self.trainer = dataset, int(batch_size), trainer
        self.generator = generator
        sampler = RandomSampler(dataset, generator=generator) if shuffle else SequentialSampler(dataset)
        self.batch_sampler = BatchSampler(sampler, self.batch_size, drop_last)

    def __len__(self) -> int:
        return len(self.batch_sampler)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        ds = self.dataset
        dev = ds.input_tensors.device
        # torch's DataLoader draws its per-epoch base seed from the loader's generator BEFORE the sampler is iterated
        # (torch/utils/data/dataloader.py, _BaseDataLoaderIter.__init__): one draw here keeps the generator's stream,
        # hence the permutation, identical to the reference's DataLoader under the same generator.
        torch.empty((), dtype=torch.int64).random_(generator=self.generator)
        for ids in self.batch_sampler:
            idx = torch.tensor(ids, dtype=torch.int64).to(dev, non_blocking=True)
            if self.trainer is not None and len(ids) == self.batch_size:
                x, y = self.trainer.input_buffers((len(ids), ds.seq_len) + tuple(ds.input_tensors.shape[1:]),
                                                  (len(ids),) + tuple(ds.output_tensors.shape[1:]))
                ds.batch_into(idx, x, y)
                yield x, y
            else:
                yield ds.batch(idx)
