"""The ``ava/`` shim of this repo overrides ONLY ``ava.models.vae``: with the repo ahead of the reference on
``sys.path`` (method 1) every other ``ava.*`` import of the example scripts
(``examples/mouse_sylls_mwe.py:20-30``) still resolves to the reference package.  Runs in a subprocess so that the
path order of this test session does not matter."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

REF = "/root/reference"

# third-party modules the reference imports at module level that are not installed here and are not on the VAE's path
STUBS = textwrap.dedent('''
    import sys, types
    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return object
    for name in ("h5py", "affinewarp", "affinewarp.crossval", "umap", "numba", "bokeh", "bokeh.plotting", "bokeh.models",
                 "bokeh.models.glyphs"):
        sys.modules.setdefault(name, _Any(name))
    sys.modules["affinewarp.crossval"].paramsearch = None
''')


def _run(code, paths):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join(paths)
    res = subprocess.run([sys.executable, "-c", code], env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return res.stdout


def test_shim_alone_provides_the_vae_module():
    out = _run("from ava.models.vae import X_SHAPE, X_DIM, VAE\n"
               "import ava\n"
               "print(VAE.__module__, X_SHAPE, X_DIM, ava.__version__)", [ROOT])
    assert out.split()[0] == "ava_amd.vae" and "(128, 128) 16384" in out


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "ava")), reason="needs the reference checkout (build container only)")
def test_shim_overrides_only_ava_models_vae():
    # the import block of examples/mouse_sylls_mwe.py:20-30, verbatim module paths
    code = STUBS + textwrap.dedent('''
        from ava.data.data_container import DataContainer
        from ava.models.vae import X_SHAPE, VAE
        from ava.models.vae_dataset import get_syllable_partition, get_syllable_data_loaders
        from ava.preprocessing.preprocess import process_sylls, tune_syll_preprocessing_params
        from ava.preprocessing.utils import get_spec
        from ava.segmenting.refine_segments import refine_segments_pre_vae
        from ava.segmenting.segment import tune_segmenting_params, segment
        from ava.segmenting.amplitude_segmentation import get_onsets_offsets
        from ava.plotting.tooltip_plot import tooltip_plot_DC
        from ava.plotting.latent_projection import latent_projection_plot_DC
        import ava, ava.models.vae, ava.models.vae_dataset, ava.data.data_container as dc
        print(VAE.__module__)
        print(ava.models.vae.__file__)
        print(ava.models.vae_dataset.__file__)
        print(dc.__file__)
        print(dc.VAE.__module__)          # DataContainer._make_latent_means (data_container.py:458-475) builds THIS class
        print(ava.__version__)
    ''')
    out = _run(code, [ROOT, REF]).split("\n")
    assert out[0] == "ava_amd.vae"
    assert out[1] == os.path.join(ROOT, "ava", "models", "vae.py")
    assert out[2].startswith(REF) and out[3].startswith(REF)
    assert out[4] == "ava_amd.vae"
    assert out[5] == "0.3.1"
