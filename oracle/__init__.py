"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU restatement of the reference algorithm for the hot path named by
BASELINE.json (per-frame Faster-RCNN feature/proposal path + SGG_emb relation head
+ instance_styleD adversarial heads).  It exists to CHECK the HIP path; nothing
under ``i2vsgg_amd/`` imports it.  Allowed importers: ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.

Layout
  c/oracle_ops.c   plain C: NMS, ROIAlign fwd/bwd, avg-pool 2x2, ROIPool fwd/bwd,
                   RPN box decode+clip (byte/integer-exact pieces)
  cops.py          ctypes binding of the C file
  rpn.py           numpy: anchors, proposal layer, anchor/proposal target layers,
                   IoU, box transforms, smooth-L1
  nets.py          torch-CPU fp32: ResNet C4 backbone (frozen BN), RPN head,
                   netD_pixel, netD_style, layer4 head, vrd relation head

Pinning (see DESIGN.md "Oracle" for the table): every function whose reference
counterpart imports in this container is checked against golden vectors produced
by that reference code (tools/gen_golden.py -> tests/golden/*.npz).  ROIAlign and
ROIPool have no runnable reference here and say "parity unpinned" in their
docstrings.
"""
