"""Mirror of the slice of rlinf/utils the token tier is called through (utils.py)."""
