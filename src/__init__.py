"""Import root used by the reference's callers (``from src.lib.SolutionsManagers import ...``)."""
