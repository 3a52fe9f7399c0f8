"""CPU oracle of the STTN inpainting hot path -- TEST INFRASTRUCTURE ONLY.

A restatement, in numpy / torch-CPU fp32, of what the reference computes on this path
(YaoFANGUK/video-subtitle-remover v1.4.0; every function cites the reference file:line it
follows).  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; the product (``video-subtitle-remover_amd/``) never
does and fails loudly when the HIP library or a GPU is missing.

Pinning status
  * network (encoder / multi-scale patch attention / FFN / decoder): PINNED -- checked against
    the reference's own ``nn.Module`` (imported from /root/reference by ``make_golden.py``)
    through the fixtures in ``tests/golden/`` (strict ``load_state_dict`` of the same weights).
  * ``batch_generator``: PINNED the same way (reference function executed by make_golden.py).
  * window schedule / u8 truncation / overlap average / wrapper loop: restated from
    ``sttn_auto_inpaint.py`` (the wrapper itself cannot be imported: it needs cv2,
    qfluentwidgets and Python >= 3.12); the reference has no test or golden vector for it.
  * cv2.resize / cv2.threshold / cv2.connectedComponentsWithStats / cv2.rectangle:
    PARITY UNPINNED -- opencv-python==4.11.0.86 (requirements.txt:2) is absent from the image
    and the mount; restated from the published OpenCV 4.11 algorithm (imgproc/resize.cpp).
  * RAFT (raft.py), flow completion (rfc.py), ProPainter generator (propainter.py): PINNED to the
    reference's modules the same way (fixtures raft.npz / rfc.npz / propainter.npz), except
    ``torchvision.ops.deform_conv2d`` (deform_conv.py): torchvision is absent, PARITY UNPINNED for
    that operator.  propainter_wrapper.py restates the plugin loop on top of them.
  * scene cuts (scene_cuts.py): PINNED to the reference's own SceneManager + ContentDetector
    (fixture scene_cuts.json: scores equal to the last bit) except ``cv2.cvtColor(BGR2HSV)``
    and ``cv2.resize`` inside it, restated from OpenCV's integer algorithms.
  * text detector (ppocr_det.py): PARITY UNPINNED -- Paddle and the weights are absent; an
    interpreter of the shipped inference programs.
"""
