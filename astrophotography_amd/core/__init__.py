"""Host-side mirrors of the reference's hot-path classes (AstroPhotography/core/__init__.py:6-19)."""
