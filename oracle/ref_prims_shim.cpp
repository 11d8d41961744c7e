// placeholder, filled below
extern "C" int ref_prims_version() { return 1; }
