"""pfpp_hip — host layer over libpfpp_hip.so (MI355X / gfx950 kernels of the PuzzleFusion++
denoise-and-verify path).  `ops` holds the tensor-level wrappers of the C ABI (include/pfpp.h);
`encoder`, `denoiser`, `verifier`, `scheduler` orchestrate them.  There is no CPU path."""
