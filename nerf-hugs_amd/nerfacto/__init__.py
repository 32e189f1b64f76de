"""Groundwork for the nerfacto path (BASELINE config 5): the encodings nerfacto takes from
tiny-cuda-nn, as HIP kernels behind the C ABI.  The nerfacto model itself (fields, proposal sampler, losses,
nerfacto/models/nerfacto.py) is not built yet."""
