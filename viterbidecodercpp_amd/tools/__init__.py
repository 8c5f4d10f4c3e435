"""Drivers around the hot path that mirror the reference's example programs (SURVEY.md section 8 f "next" rows)."""
