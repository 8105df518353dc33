# Default material (Phong)
name=default_world
diffuse_texture=assets/textures/transparency-test-diffuse-texture.png
diffuse_colorspace=srgb
diffuse_color=1,1,1,1
specular_texture=assets/textures/cobblestone_SPEC.png
specular_colorspace=linear
specular_color=1,1,1,1
norm_texture=assets/textures/cobblestone_NRM.png
normal_colorspace=linear
shininess=64
emission_color=0.0,0.0,0.0
shader=shader.default.world
pipeline=world
cutout=true
