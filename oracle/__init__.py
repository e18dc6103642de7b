"""CPU oracle for the UniGeo DepthCrafter / StableNormal hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``unigeo_amd``) may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker / the reported CPU baseline.

What it is
----------
A plain torch-CPU fp32 restatement of the algorithm behind the reference's plugin call
``DepthCrafter.forward`` (/root/reference/model/depthcrafter.py:73-99).  The arithmetic of
that call lives in third-party packages that are NOT vendored in the reference tree and
are not installed in the build container:

* huggingface ``diffusers`` (unpinned by the reference): ``UNetSpatioTemporalConditionModel``,
  ``AutoencoderKLTemporalDecoder``, ``EulerDiscreteScheduler``, ``StableVideoDiffusionPipeline``
* Tencent/DepthCrafter (only version hint: commit ee2c6e8c in a comment at
  /root/reference/model/depthcrafter.py:96): ``DepthCrafterPipeline`` and
  ``DiffusersUNetSpatioTemporalConditionModelDepthCrafter``
* ``transformers`` ``CLIPVisionModelWithProjection`` (this one IS importable here and is
  used by tests/test_oracle_clip.py to pin oracle/clip.py with random weights)

So the oracle restates the *published* SVD-XT / DepthCrafter architecture and is anchored on
the reference's call site (kwargs at model/depthcrafter.py:80-90) and on structural
known-answers (UNet parameter count 1.524 B, state-dict key set).

PARITY STATUS: **parity unpinned** at the diffusers boundary (a4-a9 of SURVEY.md section 8):
the reference holds no golden vectors or tests for that boundary and the real pipeline
cannot be run here.  The wrapper / geometry / metrics side (a2, a10, a11, a12, a13) IS pinned
by golden vectors generated from the reference's own Python (tests/golden/make_goldens.py).
"""
