from .gs_renderer import render, render_batch, render_human_scene, render_human_scene_batch  # noqa: F401
