"""MI355X-native (gfx950) SuperPoint + LightGlue hot path behind deep-image-matching's
ExtractorBase / MatcherBase plugin contracts.  See DESIGN.md."""
__version__ = "0.1.0"
