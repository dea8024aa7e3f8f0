"""CPU oracle for the decode hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import anything from this package.  The shipped path (the ``evc_amd`` package) never
imports it and fails loudly when its HIP library is missing.

What is restated here, and what pins it:

* ``schedule.py`` / ``samplers.py`` / ``scorenet.py`` / ``upfirdn2d.py`` -- the diffusion
  path (reference ``models/__init__.py``, ``models/pndm.py``,
  ``models/better/{ncsnpp_more,layerspp,layers,up_or_down_sampling}.py``,
  ``models/better/op/upfirdn2d.py``).  PINNED: ``tests/golden/make_goldens.py`` imports
  the reference itself in the build container and stores its outputs as fixtures under
  ``tests/golden/``; ``tests/test_oracle_goldens.py`` checks this restatement against them.
* ``scorenet_spade.py`` / ``unet_ddpm.py`` / ``scorenet_pseudo3d.py`` -- the alternative score networks (SPADE conditioning;
  ``models/unet.py``; the ``is3d`` / ``pseudo3d`` branches with ``models/better/layers3d.py``: archs ``unetmorepseudo3d`` and
  ``unetmore3d``).  PINNED the same way (``forward_spade.npz``, ``unet_ddpm*.npz``, ``forward_pseudo3d.npz``,
  ``forward_conv3d.npz``: outputs + forward-hook taps of every module).
* ``elic.py`` / ``entropy.py`` / ``rans.py`` -- the ELIC key-frame codec (reference
  ``Network.py``, ``ELICUtilis/layers/layers.py``, ``Inference.py``) and the entropy
  coder of third-party ``compressai==1.1.5`` (``requirements.txt:17``; not vendored, not
  installed, so the reference ELIC modules cannot be imported here).
  PARITY UNPINNED: written from the reference source text and from the published
  compressai / ryg_rans algorithm; the reference holds no golden vector for this path.
* ``exact.py`` + ``exact_conv.c`` (built by ``oracle/Makefile`` into ``oracle/_build/``) -- a
  bit-exact C restatement of the PRODUCT's fp32 convolution arithmetic (one fixed-order chain
  of fmaf per output element).  ``elic.py`` runs the entropy-parameter networks through it
  when asked for ``exact=True``: the oracle then derives bit-identical means / scales, symbols
  and bytes as the HIP codec (tests/test_gpu_elic.py, both directions, no teacher forcing).
  This pins the two implementations to each other; it does not pin either to compressai.
"""
