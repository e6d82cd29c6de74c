"""Python-side plumbing for the MI355X BWA-MEME backend: ctypes bindings of the C ABI (include/meme_hip.h),
synthetic data generators and the multi-GPU read sharding helper.  The compute path is the HIP library
(libmeme_hip.so); nothing here computes seeds or alignments on the CPU."""
