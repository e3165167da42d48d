from .map_builder import MapBuilder, SPICEComposedMapBuilder  # noqa: F401  (as euispice_coreg.synras does)
