"""TEST INFRASTRUCTURE ONLY -- runs the REFERENCE's conditioning rasteriser (pipelines.py:1200-1253, 1501-1641, 1658-1850,
1852-1902) for fixture generation and the live oracle-vs-reference CPU tests.

`/root/reference/pipelines.py` cannot be imported: its module level imports diffusers, torchvision, moviepy and three
un-vendored submodules (pipelines.py:14-27).  The rasteriser methods themselves need numpy, torch, PIL, tqdm and matplotlib
only, all of which this image has.  So the method definitions are cut out of the reference file BY NAME with `ast` at run time
and executed as methods of an empty class -- the reference's own code runs, nothing of it is stored in this repository.  Never
imported by flexam_amd, bench.py or `-m gpu` tests; only where /root/reference is mounted."""
import ast
import os

REF_ROOT = os.environ.get("FLEXAM_REFERENCE_ROOT", "/root/reference")
SOURCE = os.path.join(REF_ROOT, "pipelines.py")
METHODS = ("valid_mask", "sort_points_by_depth", "draw_rectangle", "fun_visualize_tracking_with_depth",
           "apply_cosine_positional_encoding", "_convert_frames_to_tensor", "_prepare_vis_mask", "_generate_colors_from_points",
           "_render_cosine_encoded_frame", "_visualize_cosine_encoded_tracking", "_visualize_depth_tracking", "_should_draw_point")


def reference_available() -> bool:
    return os.path.isfile(SOURCE)


def load():
    """An object carrying the reference's rasteriser methods (class FlexAMPipeline of pipelines.py, the named methods only)."""
    import matplotlib
    import numpy as np
    import torch
    from PIL import Image, ImageDraw
    from tqdm import tqdm
    src = open(SOURCE).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "FlexAMPipeline")
    found = {n.name: n for n in cls.body if isinstance(n, ast.FunctionDef)}
    missing = [m for m in METHODS if m not in found]
    if missing:
        raise RuntimeError(f"reference methods not found in {SOURCE}: {missing}")
    holder = ast.ClassDef(name="_ReferenceRasteriser", bases=[], keywords=[], body=[found[m] for m in METHODS], decorator_list=[])
    if hasattr(holder, "type_params"):
        holder.type_params = []
    mod = ast.Module(body=[holder], type_ignores=[])
    ast.fix_missing_locations(mod)
    quiet = lambda it, **kw: it                                  # the reference wraps its frame loops in tqdm progress bars
    ns = {"np": np, "torch": torch, "Image": Image, "ImageDraw": ImageDraw, "tqdm": quiet, "matplotlib": matplotlib, "os": os}
    exec(compile(mod, SOURCE, "exec"), ns)
    obj = ns["_ReferenceRasteriser"]()
    obj.output_dir = None
    return obj
