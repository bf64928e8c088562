"""Reference-side bindings over the C ABI: drop-in replacements for native modules of xinntao/EDVR (Level 2)."""
