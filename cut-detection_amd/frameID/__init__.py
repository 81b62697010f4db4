"""Drop-in `frameID` package: the reference's module and function names over the HIP path."""
