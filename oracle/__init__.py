"""CPU oracle for the UniRestore diffusion-prior restoration hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``unirestore_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / reported baseline.

What it is: a pure-torch fp32 restatement of ``DiffUIE.forward``
(/root/reference/src/modules/diffuie/unifie.py:107-169) and of every module it
reaches, with parameter names identical to the reference's checkpoints.

Parity pinning status (see DESIGN.md "Oracle"):
  * adapters written in the reference repo itself (CSCEAdapter, TaskFeatureAdapter,
    NAFBlock, AdaNAFV2, SPADE) are PINNED: tests/golden/*.npz holds inputs, weights
    and outputs produced by importing the reference classes in the build container
    (tools/gen_golden.py), and tests/test_oracle_golden.py replays them.
  * the DDIM/DDPM schedule is PINNED by closed-form known answers.
  * the diffusers-owned blocks (UNet2DConditionModel, AutoencoderKL, ResnetBlock2D,
    Transformer2DModel, Attention, schedulers; diffusers==0.29.0 per
    /root/reference/requirements.txt:14, NOT vendored, NOT installed) are restated
    from the published architecture and are "parity unpinned" beyond structure:
    parameter names + counts (865 910 724 / 83 653 863 / 52 494 080) are checked.
"""
