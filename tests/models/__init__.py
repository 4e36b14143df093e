"""model files written in the reference's style (a subclass of IonicModel whose solve() is built from tf.* calls);
they exist to exercise fib_tf_amd.traced and are not part of the product"""
