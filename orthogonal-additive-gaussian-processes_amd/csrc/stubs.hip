// (all entry points are implemented)
