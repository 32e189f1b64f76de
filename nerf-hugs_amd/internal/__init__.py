"""Host-side mirror of the reference's `MipNeRF360/internal` package for the per-ray hot path:
same module / function names, argument meaning and error behaviour, HIP kernels underneath."""
