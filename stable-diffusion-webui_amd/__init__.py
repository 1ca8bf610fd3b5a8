"""MI355X-native engine for the txt2img/img2img hot path of AUTOMATIC1111/stable-diffusion-webui."""
