"""CPU oracle for the BadDiffusion hot path -- TEST INFRASTRUCTURE ONLY.

A plain fp32 restatement (PyTorch CPU ops / numpy, no HIP, no diffusers import)
of the reference algorithm for SURVEY.md section 8 rows a-1..a-8.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import anything from this package; the product package
``baddiffusion_amd`` never does (tests/test_no_oracle_in_product.py checks it).

Parity pin: every function here is checked against golden vectors captured by
importing the reference in the build container (tests/golden/make_golden.py ->
tests/golden/*.npz) and against the known answers held by the reference's own
diffusers tests (tests/test_oracle_golden.py).  Exceptions, marked
"parity unpinned" where they occur: image-file triggers/targets (torchvision
absent in the container).
"""
