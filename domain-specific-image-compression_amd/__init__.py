"""MI355X-native hot path of the modelv2 Student-t hyperprior codec.

Host-side mirror of the reference's Python module API
(code/modelv2/{layers,distributions,model}.py,
code/modelv2/eval_selfcontained_entropy.py) over a C-ABI HIP library
(include/dsic_hip.h).  There is no CPU fallback: every compute entry point
fails loudly when the HIP library is missing.
"""
__version__ = "0.1.0"
