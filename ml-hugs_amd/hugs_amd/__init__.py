"""Host-side mirror of the reference's operator interface for the rasterizer hot path (Python, as the
reference's host side is Python).  Only what the path needs: the renderer adapter, synthetic inputs
and frame sharding.  Put the parent directory (ml-hugs_amd/) on sys.path; `import
diff_gaussian_rasterization` then resolves to the MI355X implementation, exactly as the reference's
`from diff_gaussian_rasterization import ...` expects."""
