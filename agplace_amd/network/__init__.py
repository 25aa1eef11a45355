"""Database-side ImageFE / GeM; same names as the reference's network package."""
