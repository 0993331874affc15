// empty on purpose: see ../shim.h
