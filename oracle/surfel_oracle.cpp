// placeholder, filled in below
