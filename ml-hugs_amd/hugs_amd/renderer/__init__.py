from .gs_renderer import render, render_human_scene  # noqa: F401
