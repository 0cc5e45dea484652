"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy / plain C / torch-CPU fp32) of the EAS-SNN hot path
(SURVEY.md section 8a rows a1..a14).  It exists to CHECK the HIP path, never
to BE it: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import anything from this package.  The product
package (``eas_snn_amd``) never imports it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * events_ref (a1..a4)          pinned: golden vectors produced by the real
                                 reference code (tests/golden/events_*.npz).
  * embedding_ref (a5..a7)       pinned: golden vectors produced by the real
                                 reference ``yolox/models/embedding.py``.
  * sj_ref (a9..a11, a14)        PARITY UNPINNED at the spikingjelly boundary:
                                 spikingjelly==0.0.0.0.14 is not vendored in
                                 /root/reference and is not installable here.
                                 The restatement follows the published
                                 semantics and is cross-checked against the
                                 reference's in-repo ``LIFLayer``/``LIFCell``
                                 (same recurrence except '>' vs '>=').
  * model_ref (a8, a12, a13)     pinned on topology: golden logits/losses from
                                 the unmodified reference model classes built
                                 on top of sj_ref (the only stand-in).
"""
