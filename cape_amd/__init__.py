"""cape_amd -- MI355X (gfx950) native implementation of CAPE's Chebyshev mesh-convolution hot
path behind the reference's ``lib.models.CAPE`` API.

Sub-modules:
  mesh_sampling, load_data  host-side operator precompute / loading (scipy; no GPU needed)
  graph                     operator algebra: precomposed CSR operators per layer
  ops                       autograd operators calling libcape_hip.so (needs the built library)
  models                    ``CAPE`` class: build_graph / fit / encode / decode / predict ...
  dist                      data-parallel helpers (flat gradient bucket, RCCL all-reduce)
The compute modules raise at import if libcape_hip.so is missing: there is no CPU fallback.
"""
__version__ = "0.1.0"
