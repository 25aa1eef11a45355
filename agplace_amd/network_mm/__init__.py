"""Query-side modules; same module/class names as the reference's network_mm package."""
