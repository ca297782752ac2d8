"""pfpp_hip — host layer over libpfpp_hip.so (MI355X / gfx950 kernels of the PuzzleFusion++
denoise-and-verify path).  `ops` holds the tensor-level wrappers of the C ABI (include/pfpp.h);
`encoder`, `denoiser`, `verifier`, `scheduler` orchestrate them.  There is no CPU path."""

import os as _os

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order; two streams on one queue run one after the
# other.  The training schedule has four streams that must run side by side (dependency chain, weight gradients, next batch's encoder,
# the caller's default stream); in a process that has created other streams before (torch hands out pooled streams round-robin) two of
# them can land on one queue — measured: the module-surface iteration 6.2 -> 10.8 ms behind 290 other tests in one process.  Eight queues
# keep them apart; the variable is read when the HIP runtime initialises, so it is set before the first use of the device.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
