"""MI355X-native SmartVidCrop saliency-to-crop hot path (drop-in for bmezaris/RetargetVid's
smartVidCrop.py entry points).  See DESIGN.md and INTEGRATION.md."""
__version__ = '0.1.0'
