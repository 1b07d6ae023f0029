"""CPU oracle for the TextFlux / FLUX.1-Fill denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / reported CPU baseline.  The
product path (``textflux_amd``) never imports this package and raises if the HIP
library is missing.

Parity status: **pinned** against the reference itself -- every function here is
checked (tests/test_oracle_golden.py) against golden vectors produced by
importing the reference (`/root/reference/diffusers/src`, diffusers 0.32.0.dev0)
in the build container with ``tests/golden/make_goldens.py``.  Two sub-paths of
the reference's call chain live in third-party packages that are absent from
/root/reference and are therefore *unpinned* (see DESIGN.md): the CLIP/T5 text
encoders (`transformers==4.43.3`) and PEFT's LoRA forward (`peft`, unpinned).
"""
